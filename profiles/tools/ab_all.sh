# same-box A/B of two full builds (libsxgpoa_devA.so vs libsxgpoa_devC.so): parity suite on C first, then headline and workloads
cd $GRAFT_REPO_ROOT
if [ -n "$PARITY" ]; then SXG_POA_LIB=$GRAFT_REPO_ROOT/smoothxg_amd/csrc/libsxgpoa_devC.so timeout 1500 python -m pytest tests -m gpu -x -q $PARITY 2>&1 | tail -5; fi
bash profiles/tools/ab.sh
[ -n "$WLS" ] && bash profiles/tools/ab_wl.sh "$WLS"
