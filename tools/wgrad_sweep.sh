set -e
cd $GRAFT_REPO_ROOT
for b in 256 512 768 1024; do
  echo "== PN2_WGRAD_BLOCKS=$b"
  PN2_WGRAD_BLOCKS=$b timeout -k 10 200 python tools/kernel_table.py 2>/dev/null | grep -E "sum of|gemm_wgrad  *x  5|slab_reduce  *x  5"
done
