"""Per-kernel SQ counter summary from two rocprofv3 --pmc passes (pass A: SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; pass B: SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE).  Groups dispatches by (kernel name, grid, workgroup): each conv layer shape is one row.
mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / 32 (the busy counter is per shader engine, 32 of them; the MFMA counter per
SIMD, 1024), lds_act% = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES / 8 (per CU, 256); bankconf% = conflict cycles / LDS active cycles."""
import csv, glob, sys, collections

def load(d):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aic::", "").replace("(anonymous namespace)::", "")[:60], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    out = {}
    for k, v in rows.items():
        n = max(cnt[(k, c)] for c in v)
        out[k] = {c: x / n for c, x in v.items()}
        out[k]["_n"] = n
    return out

a = load(sys.argv[1]); b = load(sys.argv[2])
keys = sorted(a, key=lambda k: -a[k].get("SQ_BUSY_CYCLES", 0) * a[k]["_n"])
print(f"{'kernel':48s} {'grid':>8s} {'wg':>4s} {'n':>4s} | {'mfma_busy%':>9s} {'valu/mfma':>9s} {'lds/mfma':>8s} {'salu/mfma':>9s} {'vmem/mfma':>9s} {'wait_any%':>9s} {'wait_inst%':>10s} {'wait_lds%':>9s} {'bankconf%':>9s} {'lds_act%':>8s}")
for k in keys[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    x, y = a[k], b.get(k, {})
    mf = max(x.get("SQ_INSTS_MFMA", 0), 1)
    wc = max(x.get("SQ_WAVE_CYCLES", 0), 1)
    busy = max(x.get("SQ_BUSY_CYCLES", 0), 1)
    print(f"{k[0][:48]:48s} {k[1]:8d} {k[2]:4d} {int(x['_n']):4d} | {100*x.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/busy/32:9.1f} {x.get('SQ_INSTS_VALU',0)/mf:9.2f} {y.get('SQ_INSTS_LDS',0)/mf:8.2f} {y.get('SQ_INSTS_SALU',0)/mf:9.2f} {y.get('SQ_INSTS_VMEM',0)/mf:9.2f} {100*x.get('SQ_WAIT_ANY',0)/wc:9.1f} {100*x.get('SQ_WAIT_INST_ANY',0)/wc:10.1f} {100*y.get('SQ_WAIT_INST_LDS',0)/wc:9.1f} {100*y.get('SQ_LDS_BANK_CONFLICT',0)/max(y.get('SQ_LDS_IDX_ACTIVE',1),1):9.1f} {100*y.get('SQ_LDS_IDX_ACTIVE',0)/busy/8:8.1f}")

# The conv class as bench.py defines it (every MFMA conv kernel except the two fused 3-channel stems), weighted by busy cycles
# = by time: the figure to hold against the north star's "MFMA utilisation on the conv kernels".
def conv_class(name):
    return any(t in name for t in ("conv_igemm", "conv3x3", "conv1x1", "c2f16"))

num = sum(a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * a[k]["_n"] for k in a if conv_class(k[0]))
den = sum(a[k].get("SQ_BUSY_CYCLES", 0) * a[k]["_n"] for k in a if conv_class(k[0]))
nl = sum(int(a[k]["_n"]) for k in a if conv_class(k[0]))
reid = lambda k: conv_class(k[0]) and ("c64_block" in k[0] or "pp_patch" in k[0] or "sp_patch" in k[0] or "igemm_pp" in k[0])
num_r = sum(a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * a[k]["_n"] for k in a if reid(k))
den_r = sum(a[k].get("SQ_BUSY_CYCLES", 0) * a[k]["_n"] for k in a if reid(k))
if den:
    print(f"\nconv class, time-weighted over {nl} launches: MFMA busy {100 * num / den / 32:.1f} % of the busy cycles "
          f"(ReID trunk kernels c64_block / pp_patch / sp_patch / s2_sp_patch / igemm_pp alone: {100 * num_r / max(den_r, 1) / 32:.1f} %, "
          f"{100 * den_r / den:.0f} % of the class's time)")
