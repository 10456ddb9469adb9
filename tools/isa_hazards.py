"""CLI shim: the scanner lives in rba_amd/csrc/isa_hazards.py since round 6 (the build runs it as a post-link gate).
Usage: python tools/isa_hazards.py [lib.so | file.o ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rba_amd.csrc.isa_hazards import *            # noqa: F401,F403,E402
from rba_amd.csrc.isa_hazards import LLVM, main   # noqa: F401,E402

if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
