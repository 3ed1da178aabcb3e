cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
pt() { timeout 200 python tools/ab_loop.py --label new --points "$1" --steps 96 --rounds 1 2>&1 | grep "^round\|Error\|error" | head -5; }
for r in 1 2 3; do pt 12:3; pt 12:2; pt 8:3; pt 8:2; pt 14:3; pt 10:3; done 2>&1 | tee gpurun_out/ab_s27_fused.txt
