cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== statistics behind the upload (default)"; timeout -k 10 200 python3 tools/host_path_time.py 2>/dev/null
echo "== SAPCA_UPLOAD_STATS_OFF=1"; SAPCA_UPLOAD_STATS_OFF=1 timeout -k 10 200 python3 tools/host_path_time.py 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_up -o up -- python3 tools/host_path_time.py > gpurun_out/prof_up.log 2>&1
python3 - <<'PY'
import csv,glob
fs=glob.glob('gpurun_out/prof_up/**/*kernel_stats.csv',recursive=True)
if fs:
    for r in list(csv.DictReader(open(fs[0]))):
        if 'colstats' in r['Name'] or 'max_exponent' in r['Name'] or 'at_stats' in r['Name'] or 'tile_index' in r['Name']:
            print(r['Name'][:80], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
rm -rf gpurun_out/prof_up
