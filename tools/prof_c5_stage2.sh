cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/pc5 -o c5 --output-format csv -- python3 tools/stage2_sorted_ab.py c5 > gpurun_out/pc5.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pc5/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0][:70]
        print(f'{n:72s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:10.1f} us  total {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
