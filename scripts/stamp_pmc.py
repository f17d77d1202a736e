"""Fill profiles/pmc_traffic.json's stamp with the git commit whose sources the counters were taken from: run HERE (where .git
exists) after copying the file back from a gpurun call.  The sources are identified by stamp.source_sha (pwcnet_amd.profiler.
source_stamp); this script only records which commit holds exactly those sources (HEAD if the tree's hash matches, else none)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pwcnet_amd.profiler import source_stamp
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
t = json.load(open(path))
now = source_stamp()
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "pwcnet_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
for st in (t.get("stamp"), (t.get("op_leg") or {}).get("stamp")):
    if st is None:
        continue
    if st.get("source_sha") == now and not dirty:
        st["git_sha"] = head
    print("stamp", st.get("source_sha"), "tree", now, "git_sha", st.get("git_sha"))
json.dump(t, open(path, "w"), indent=1)
