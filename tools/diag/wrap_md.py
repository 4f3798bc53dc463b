#!/usr/bin/env python3
"""Re-wraps the prose of a markdown file at 120 columns (tables, headings, code fences and their contents are left as they are;
list items keep a hanging indent).  usage: tools/diag/wrap_md.py FILE..."""
import re
import sys
import textwrap

WIDTH = 120


def wrap_file(path):
    out, para, in_code = [], [], False
    lines = open(path).read().split("\n")

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)((?:[*\-+]|\d+\.)\s+)?", first)
        lead, bullet = m.group(1), m.group(2) or ""
        text = " ".join([first[len(lead) + len(bullet):].strip()] + [l.strip() for l in para[1:]])
        out.extend(textwrap.wrap(text, WIDTH, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet),
                                 break_long_words=False, break_on_hyphens=False) or [lead + bullet])
        para.clear()
    for line in lines:
        if line.strip().startswith("```"):
            flush()
            in_code = not in_code
            out.append(line)
            continue
        if in_code or line.startswith("|") or line.startswith("#") or not line.strip() or line.startswith("<") or line.startswith("{"):
            flush()
            out.append(line)
            continue
        starts_item = re.match(r"^\s*(?:[*\-+]|\d+\.)\s+", line) is not None
        if starts_item:
            flush()
        para.append(line)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        wrap_file(p)
