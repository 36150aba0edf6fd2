cd $GRAFT_REPO_ROOT
for mode in "" "--det"; do for tile in 42 22 41; do
echo "=== tile cap $tile mode '$mode'"
Y5_WG3_TILE=$tile python scripts/wgrad_bench.py --k3-ab --cfgs $([ $tile = 42 ] && echo 1,3 || echo 3) --iters 6 $mode 2>&1 | grep -v amdgpu.ids
done; done
