"""SHA-256 over the device-code sources of the library (csrc/*.hip, *.hip.h, *.h, *.inc, build.sh; not the host-only *.cpp): how a measurement file
(profiles/pmc_traffic.json) names the build it was taken on and how bench.py notices that the tree has moved on."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_sources_sha256(csrc=CSRC):
    files = sorted(p for pat in ("*.hip", "*.h", "*.inc", "build.sh") for p in glob.glob(os.path.join(csrc, pat))
                   if os.path.basename(p) != "host_internal.h")       # (included by the host-only *.cpp alone)
    h = hashlib.sha256()
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha256())
