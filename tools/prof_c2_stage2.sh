cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/pc2 -o c2 --output-format csv -- python3 tools/stage2_sorted_ab.py c2 > gpurun_out/pc2.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pc2/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:64]
        if any(k in n for k in ("sig", "fix_sort", "export")):
            print(f'{n:66s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
rm -rf gpurun_out/pc2
