"""sha256 over the sources the shared library is built from (markovflow_amd/csrc/*.hip, *.hpp, Makefile and the C header).
profiles/hbm_traffic.json is keyed on it: bench.py quotes a PMC traffic figure only when it was collected on exactly these
sources (mf_version() went unbumped through eight library-changing commits in round 2).   python3 scripts/csrc_hash.py"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root: str = ROOT) -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "markovflow_amd", "csrc", "*.h*")) + [os.path.join(root, "markovflow_amd", "csrc", "Makefile"),
                                                                                      os.path.join(root, "include", "markovflow_amd.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
