import argparse
import os
import sys

from .generator import generate_all

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Generate the op namespaces from the DSL definitions")
    ap.add_argument("--namespaces", default="", help="comma separated subset (default: all)")
    ap.add_argument("--check", action="store_true", help="do not write; exit 1 if a checked-in file is stale")
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    stale = generate_all(root, [n for n in a.namespaces.split(",") if n] or None, a.check)
    for p in stale:
        print(("stale: " if a.check else "wrote: ") + os.path.relpath(p, root))
    sys.exit(1 if (a.check and stale) else 0)
