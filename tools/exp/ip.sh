set -euo pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/infer_prof2
rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 tools/infer_gaps.py > /dev/null 2>&1
python3 tools/infer_gaps.py --digest $O/prof | cut -c1-60,100-160
rm -rf $O/prof
