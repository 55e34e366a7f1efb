"""SHA-1 over the sources libhavc_mi355.so is built from (every *.hip / *.cpp / *.h / *.inc under vsdeoldify_amd/csrc except the generated build_stamp.h, plus
include/havc_mi355.h and the Makefile), in sorted order, each as `name NUL content NUL`.  The Makefile writes it into csrc/build_stamp.h, the library exports it as
havc_build_stamp(), and tests/test_host_logic.py compares it with the tree: a shipped .so that does not match its sources (the binary is git-ignored
and travels prebuilt) fails a CPU test wherever the suite runs.
   python tools/build_stamp.py          -> prints the header line
   python tools/build_stamp.py --hash   -> prints the hash"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vsdeoldify_amd", "csrc")


def source_files():
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".inc")) and f != "build_stamp.h")
    return [os.path.join(CSRC, f) for f in names] + [os.path.join(CSRC, "Makefile"), os.path.join(ROOT, "include", "havc_mi355.h")]


def stamp():
    h = hashlib.sha1()
    for p in source_files():
        h.update(os.path.basename(p).encode() + b"\0" + open(p, "rb").read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(stamp() if "--hash" in sys.argv else f'#define HAVC_BUILD_STAMP "{stamp()}"')
