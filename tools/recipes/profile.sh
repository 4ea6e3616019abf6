set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile.sh ${1:-r02} > gpurun_out/profile_${1:-r02}.log 2>&1
python3 tools/summarize_profile.py ${1:-r02} > gpurun_out/profile_${1:-r02}_summary.log 2>&1
tail -5 gpurun_out/profile_${1:-r02}_summary.log; ls profiles/ | head -30; cp profiles/${1:-r02}_* gpurun_out/ 2>/dev/null
head -12 gpurun_out/${1:-r02}_kernel_stats.csv
