#!/usr/bin/env python3
"""Takes one summary of scripts/pmc_summary.py (rocprofv3 --pmc passes of scripts/pmc_collect.sh on ONE workload) into
profiles/pmc_traffic.json: HBM-side bytes per working launch (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM) per kernel
class of bench.py, and WITH them what they were measured on -- the demangled kernel names, the kernel family of the fused multiplies,
the git revision, the round -- so that bench.py can tell a kept figure from a figure of a kernel that no longer runs.
usage: python3 scripts/pmc_to_traffic.py <workload> <summary.json> <round tag> [git sha]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, summary, tag = sys.argv[1], json.load(open(sys.argv[2])), sys.argv[3]
sha = sys.argv[4] if len(sys.argv) > 4 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
# where the epilogue number sits among the template arguments of each multiply family (tfq_spmm.hip), and where FIRST (first-iteration instance)
EPI_AT = {"k_spmm_ilv16": 0, "k_spmm_ilv16f": 0, "k_spmm_ilv8": 0, "k_spmm_ilv8b": 0, "k_spmm_ilvf": 2, "k_spmm_ilv8w": 1, "k_spmm_ilv8f": 1,
          "k_spmm_mfma": 3, "k_spmm_mfma8": 3, "k_spmm_small4": 2, "k_spmm_m4": 2, "k_spmm_s4w": 2, "k_spmm_direct": 3}
FIRST_LAST = {"k_spmm_ilv16", "k_spmm_ilv16f", "k_spmm_ilv8", "k_spmm_ilv8b", "k_spmm_ilvf", "k_spmm_ilv8w", "k_spmm_ilv8f"}
VEC = {"k_xpay_v6": "xpay_v6", "k_v5_nrm": "v5_nrm", "k_x_v6_v7": "x_v6_v7"}
entry, names = {}, {}
def total(v): return int(round((v["hbm_read_MB(2x FETCH_SIZE)"] + v["hbm_write_MB"]) * 1e6))
spmm = []
for name, v in summary.items():
    if "hbm_read_MB(2x FETCH_SIZE)" not in v or not v.get("working_launches"): continue
    m = re.match(r"(k_\w+)<(.*)>\(", name)
    if not m: continue
    fam, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if fam in VEC:
        entry[VEC[fam]] = total(v); names[VEC[fam]] = name.split("(")[0]
    elif fam in EPI_AT:
        if fam in FIRST_LAST and args[-1] == "true" and len(args) > EPI_AT[fam] + 1: continue     # the first-iteration instance
        spmm.append((fam, int(args[EPI_AT[fam]]), name.split("(")[0], v))
fused = [f for f in spmm if f[1] in (1, 2)]
family = fused[0][0] if fused else None
for fam, epi, nm, v in spmm:
    key = {1: "spmm_v4_dot", 2: "spmm_v5_nrm_dot"}.get(epi) if fam == family else None
    if epi == 0: key = "multiply" if fam == family else "multiply_native_api"
    if key and (key not in entry or v["working_launches"] > 0): entry[key] = total(v); names[key] = nm
entry["_kernel"], entry["_kernels"], entry["_sha"], entry["_round"] = family, names, sha, tag
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d["_note"] = ("HBM-side bytes per working launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/pmc_collect.sh; FETCH_SIZE doubled as "
              "/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950), one entry per workload written by scripts/pmc_to_traffic.py: `_kernel` is the kernel "
              "family of the fused multiplies the figures were taken on, `_kernels` the demangled names per class, `_sha` the revision, `_round` the round; bench.py quotes an "
              "entry only while tfqmrgpuExt_getMultiplyKernel names the same family for the running plan")
d[wl] = entry
json.dump(d, open(path, "w"), indent=1)
print(wl, json.dumps({k: v for k, v in entry.items() if not k.startswith("_kernels")}))
