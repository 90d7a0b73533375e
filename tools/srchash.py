"""A hash of the kernel and host sources of libgamdp (gam_ngs_amd/csrc + include/), computable without git: the GPU box has
no .git.  tools/summarise_profiles.py records it next to every profile set; bench.py compares it with the tree it runs
from and says in its line whether the replayed counters belong to the source the run uses ("profile_matches_source")."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(root=ROOT):
    h = hashlib.sha256()
    files = []
    for d in ("gam_ngs_amd/csrc", "include"):
        for name in sorted(os.listdir(os.path.join(root, d))):
            if name.endswith((".hip", ".inc", ".h", ".cpp")) or name == "Makefile":
                files.append(os.path.join(d, name))
    for rel in files:
        h.update(rel.encode() + b"\0")
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
