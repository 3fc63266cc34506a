"""MFMA utilisation of one bench step from a rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o m -- \
        python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline
    python tools/mfma_util.py gpurun_out/pmc_mfma/m_counter_collection.csv profiles/r02_mfma_util.json

MfmaUtil (rocprofiler-sdk counter_defs.yaml, gfx950) = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100 with
SIMD_NUM = 256 CUs x 4; rocprofv3's CSV holds the SUM of GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so the
per-dispatch active-cycle count is that sum / 8.  Reported: per kernel family, and for the whole step relative to the time the GPU is
busy (the sum over dispatches; under the counter pass kernels run serialised, so this equals the step's GPU time)."""
import collections, csv, hashlib, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "uni-encoder-code_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            with open(os.path.join(d, n), "rb") as f:
                h.update(n.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    except Exception:
        return "?"


if __name__ == "__main__":
    per = collections.defaultdict(dict)
    name = {}
    for r in csv.DictReader(open(sys.argv[1])):
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    ids = sorted(per, key=int)
    marks = [i for i in ids if "cast_multi" in name[i]]
    seg = [i for i in ids if int(marks[-2]) <= int(i) < int(marks[-1])] if len(marks) >= 2 else ids     # one full step
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for i in seg:
        c = per[i]
        f = re.sub(r"<.*", "", name[i])
        fam[f][0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); fam[f][1] += c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0; fam[f][2] += 1
    mf = sum(v[0] for v in fam.values()); act = sum(v[1] for v in fam.values())
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline (one full step)",
           "formula": "sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", "git_head": git_head(), "kernel_sources_sha": sources_sha(),
           "mfma_busy_frac_of_gpu_busy": round(mf / (act * 1024.0), 4), "dispatches": len(seg),
           "families": {f: {"launches": v[2], "mfma_util": round(v[0] / (v[1] * 1024.0), 4) if v[1] else 0.0, "share_of_active_cycles": round(v[1] / act, 4)}
                        for f, v in sorted(fam.items(), key=lambda x: -x[1][1])[:24]}}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
