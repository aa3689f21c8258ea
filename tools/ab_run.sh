# same-box A/B of two builds of the library on any command:  bash tools/ab_run.sh <a.so> <b.so> <rounds> <command ...>
R=$GRAFT_REPO_ROOT
L=$R/selfsupervised-nvsf_amd/lib/libnvsf_hip.so
A=$1; B=$2; N=$3; shift 3
for i in $(seq 1 $N); do
  for v in $A $B; do
    cp $R/$v $L
    echo "== $v"
    "$@" 2>&1 | tail -${TAIL:-3}
  done
done
