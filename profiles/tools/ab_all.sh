# same-box A/B of two full builds (libsxgpoa_devA.so vs libsxgpoa_devC.so): parity on C first (PARITY = pytest arguments), then
# the headline (ab.sh) and the workloads in WLS (ab_wl.sh)
cd $GRAFT_REPO_ROOT
if [ -n "$PARITY" ]; then SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_devC.so timeout 1500 python -m pytest $PARITY -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5; fi
bash profiles/tools/ab.sh
[ -n "$WLS" ] && bash profiles/tools/ab_wl.sh "$WLS"
