# usage (GPU box): bash tools/prep_time.sh "base fe1 ..."  -- per-kernel time of the format builders under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $1; do
  if [ "$v" != "default" ]; then export SAPCA_LIB_PATH=single-algebra_amd/lib/exp/libsapca_$v.so; fi
  rm -rf gpurun_out/prof_prep_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_prep_$v -- python tools/abl_run.py > /dev/null 2>&1 || exit 1
  python - <<P
import csv,glob
import os; f=max(glob.glob("gpurun_out/prof_prep_$v/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if 'sapca' in r['Name'] and float(r['AverageNs']) > 60000:
        n=r["Name"]; i=n.find("sapca::k::"); print("$v", n[i+33:i+70].ljust(40), r['Calls'], round(float(r['AverageNs'])/1e3,1))
P
done
