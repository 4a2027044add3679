# needs vggsfm_amd/_variants/lib_head.so: scripts/prof/build_ref_variant.sh <git-ref of the build to compare with>
# c5 video, in-tree library against the previous build (same box)
for v in head new head new; do
  if [ $v = head ]; then export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_head.so; else unset VGGSFM_AMD_LIB; fi
  python scripts/run_c5_video.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5', round(d['total_seconds'],3), round(d['joint_ba_seconds_total'],3), round(d['final_joint_problem_iteration_ms'],3), {k:round(x,3) for k,x in d['final_joint_problem_kernel_ms'].items()})"
done
