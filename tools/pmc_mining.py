#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/mine_once.py -> HBM-side bytes per pool sweep.

  python tools/pmc_mining.py FETCH_DIR WRITE_DIR OUT.json SWEEPS

Only the library's own kernels are counted (the scorer's contraction launches and the selection kernels); torch's
kernels that generate the synthetic pool are left out.  Corrections as in tools/pmc_summary.py: FETCH_SIZE doubled,
WRITE_SIZE as read, both in KiB."""
import collections, csv, glob, json, re, sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [set(), 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)
        per[n][0].add(r["Dispatch_Id"])
        per[n][1] += float(r["Counter_Value"])
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
sweeps = int(sys.argv[4])
mine = [n for n in set(fetch) | set(write) if n.startswith(("void gemm_", "gemm_", "topk_", "void topk_", "void pool_stream_", "pool_stream_"))]
per = {}
tot_r = tot_w = 0.0
for n in sorted(mine):
    rd, wr = 2.0 * fetch[n][1] * 1024, write[n][1] * 1024
    per[n] = {"launches_per_sweep": max(len(fetch[n][0]), len(write[n][0])) / sweeps, "read_bytes_per_sweep": rd / sweeps,
              "write_bytes_per_sweep": wr / sweeps}
    tot_r += rd; tot_w += wr
out = {"sweep": {"hbm_bytes_per_sweep": (tot_r + tot_w) / sweeps, "read_bytes_per_sweep": tot_r / sweeps, "write_bytes_per_sweep": tot_w / sweeps,
                 "sweeps_measured": sweeps},
       "per_kernel": per,
       "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as read; KiB -> bytes; library kernels only"}
# digest of the kernel sources the counters were taken with (bench.py withholds the value once they have changed)
import hashlib, os  # noqa: E401,E402
_d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pacednegatives_amd", "csrc")
_h = hashlib.sha256()
for _f in sorted(os.listdir(_d)):
    with open(os.path.join(_d, _f), "rb") as _fh:
        _h.update(_f.encode() + b"\0" + _fh.read())
out["kernel_sources_sha256"] = _h.hexdigest()
json.dump(out, open(sys.argv[3], "w"), indent=1)
for n, v in sorted(per.items(), key=lambda kv: -kv[1]["read_bytes_per_sweep"]):
    print(f"{n[:70]:70s} launches/sweep={v['launches_per_sweep']:7.1f} rd={v['read_bytes_per_sweep']/1e9:8.3f} GB wr={v['write_bytes_per_sweep']/1e9:8.3f} GB")
print("per sweep:", out["sweep"])
