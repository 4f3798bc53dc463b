"""bindings/rust/ has never met a compiler (no rustc in the image), so this CPU test guards it mechanically: the raw
binding `src/sys.rs` must declare exactly the functions of `include/jpegenc_mi355x.h`, with the same number of
arguments and matching types, its `#[repr(C)]` structs must list the header's fields in the header's order with the same
widths, its constants must carry the header's values, and every `sys::` item the safe layer (`src/lib.rs`) uses must
exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "jpegenc_mi355x.h")
SYS_RS = os.path.join(ROOT, "bindings", "rust", "src", "sys.rs")
LIB_RS = os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")

C_SCALARS = {"int": "c_int", "int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "int16_t": "i16", "uint8_t": "u8",
             "uint64_t": "u64", "size_t": "usize", "char": "c_char", "void": "c_void"}


def strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def split_top(s, sep=","):
    """split at separators that are not inside brackets"""
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "([{"
        depth -= ch in ")]}"
        if ch == sep and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_decl(decl, field):
    """One C declarator -> (rust type, name).  `const uint8_t *const *frames` -> `*const *const u8`; `T x[N]` is
    `[T; N]` in a struct and a pointer to T as a parameter."""
    m = re.match(r"^(.*?)\(\s*\*\s*(\w+)\s*\)\s*\[(\w+)\]$", decl.strip())       # pointer to an array: `const T (*name)[N]`
    if m:
        inner, _ = c_decl(m.group(1) + " x", field)
        return ("*const " if "const" in m.group(1).split() else "*mut ") + f"[{inner}; {m.group(3)}]", m.group(2)
    decl = " ".join(decl.replace("*", " * ").split())
    dims = re.findall(r"\[(\w*)\]", decl)
    decl = re.sub(r"\s*\[\w*\]", "", decl)
    tokens = decl.split()
    name = ""
    if tokens[-1] not in ("*", "const") and len([t for t in tokens if t not in ("*", "const", "struct")]) > 1:
        name = tokens.pop()
    base = next(t for t in tokens if t not in ("*", "const", "struct"))
    rust = C_SCALARS.get(base, base)
    # walk the pointer levels from the base type outwards; a level is *const when the thing it points to is const
    segments = " ".join(tokens).split("*")
    for level in range(len(segments) - 1):
        rust = ("*const " if "const" in segments[level].split() else "*mut ") + rust
    base_const = "const" in segments[0].split()
    for n in reversed(dims):
        rust = f"[{rust}; {n}]" if field else ("*const " if base_const else "*mut ") + rust
        if not field:
            break                                                    # (only the outermost dimension decays)
    return rust, name


def parse_header():
    raw = strip_c_comments(open(HEADER).read())
    text = re.sub(r"^\s*#[^\n]*", " ", raw, flags=re.M)
    text = text.replace('extern "C" {', " ")
    structs, enums = {}, {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for stmt in (s.strip() for s in m.group(2).split(";")):
            if not stmt:
                continue
            parts = split_top(stmt)
            spec = re.match(r"^(.*?)[\w\[\]]+$", parts[0]).group(1)            # `int32_t h[4], v[4]` shares its specifier
            for k, d in enumerate(parts):
                rust, name = c_decl(d if k == 0 else spec + d, field=True)
                fields.append((name, rust))
        structs[m.group(1)] = fields
    for m in re.finditer(r"typedef\s+enum\s*\w*\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        value = -1
        for item in split_top(m.group(1)):
            if "=" in item:
                key, expr = (x.strip() for x in item.split("=", 1))
                value = eval(expr, {"__builtins__": {}}, dict(enums))              # integer expressions of literals / earlier enumerators
            else:
                key, value = item.strip(), value + 1
            enums[key] = value
    text = re.sub(r"typedef\s+(struct|enum)\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", " ", text)
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(jpegenc_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [c_decl(a, field=False) for a in split_top(args)]
        funcs[name] = (None if ret == "void" else c_decl(ret + " r", field=False)[0], params)
    return funcs, structs, enums


def parse_sys_rs():
    text = re.sub(r"//[^\n]*", " ", open(SYS_RS).read())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        structs[m.group(1)] = [(f.split(":")[0].replace("pub", "").strip(), " ".join(f.split(":", 1)[1].split()))
                               for f in split_top(m.group(2)) if ":" in f]
    funcs = {}
    ext = re.search(r'extern\s+"C"\s*\{(.*)\}', text, flags=re.S).group(1)
    for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        params = [(" ".join(a.split(":", 1)[1].split()), a.split(":")[0].strip()) for a in split_top(m.group(2)) if ":" in a]
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, params)
    consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub\s+const\s+(\w+)\s*:\s*\w+\s*=\s*(-?\w+)\s*;", text)}
    return funcs, structs, consts


def same_type(c_rust, rs):
    rs = rs.replace("core::ffi::", "")
    if c_rust == rs:
        return True
    if re.sub(r"\[(\w+); \w+\]", r"\1", rs) == c_rust:      # `const T tables[2]` may be bound as *const [T; 2]: same address, same layout
        return True
    return c_rust in ("hipStream_t", "*mut c_void") and rs == "*mut c_void"


def test_sys_rs_declares_exactly_the_header():
    h_funcs, h_structs, h_enums = parse_header()
    r_funcs, r_structs, r_consts = parse_sys_rs()
    assert len(h_funcs) >= 60 and len(h_structs) >= 5
    assert set(h_funcs) == set(r_funcs), (sorted(set(h_funcs) - set(r_funcs)), sorted(set(r_funcs) - set(h_funcs)))
    for name, (ret, params) in h_funcs.items():
        r_ret, r_params = r_funcs[name]
        assert len(params) == len(r_params), (name, params, r_params)
        assert (ret is None) == (r_ret is None) and (ret is None or same_type(ret, r_ret)), (name, ret, r_ret)
        for (ct, cn), (rt, rn) in zip(params, r_params):
            assert same_type(ct, rt), (name, cn, ct, rt)
    for name, fields in h_structs.items():
        assert name in r_structs, name
        assert [f for f, _ in fields] == [f for f, _ in r_structs[name]], (name, fields, r_structs[name])
        for (fn, ct), (_, rt) in zip(fields, r_structs[name]):
            assert same_type(ct, rt), (name, fn, ct, rt)
    assert len(r_consts) >= 10
    for key, value in r_consts.items():
        assert h_enums.get(key) == value, (key, value, h_enums.get(key))


def test_lib_rs_only_uses_what_sys_rs_declares():
    r_funcs, r_structs, r_consts = parse_sys_rs()
    sys_text = open(SYS_RS).read()
    declared = set(r_funcs) | set(r_structs) | set(r_consts) | set(re.findall(r"pub\s+(?:type|enum)\s+(\w+)", sys_text))
    used = set(re.findall(r"\bsys::(\w+)", open(LIB_RS).read()))
    assert used, "lib.rs is expected to go through sys::"
    assert used <= declared, sorted(used - declared)
