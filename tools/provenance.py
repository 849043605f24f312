"""Which code a profile was taken from: the git HEAD (CDF_GIT_HEAD in the environment on the GPU box, which has no .git; else
`git rev-parse HEAD`) and a digest of the kernel sources (csrc/*.hip, csrc/*.h, include/*.h).  tools/prof_summary.py and
tools/pmc_traffic.py stamp it into the JSON summaries they write; bench.py only quotes numbers from a committed profile whose
`csrc_sha16` equals that of the sources it runs (`roofline.traffic_provenance`)."""
import glob
import hashlib
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, "cold-diffusion-models_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(REPO, "cold-diffusion-models_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(REPO, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def git_head():
    head = os.environ.get("CDF_GIT_HEAD", "").strip()
    if head:
        return head
    try:
        return subprocess.run(["git", "-C", REPO, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def stamp():
    return {"git_head": git_head(), "csrc_sha16": csrc_sha16()}
