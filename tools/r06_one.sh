# bash tools/r06_one.sh <log tag> <pytest args...>
cd $GRAFT_REPO_ROOT
T=$1; shift
timeout -k 10 900 python -m pytest "$@" > gpurun_out/r06_$T.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r06_$T.log | cut -c1-500
