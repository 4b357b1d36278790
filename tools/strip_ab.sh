# A/B of compile-time variants of the strip kernels on ONE box: builds libm2h with each flag set, runs tools/strip_bench.py on each twice (interleaved).
cd $GRAFT_REPO_ROOT
C=move2hear-active-av-separation_amd/csrc
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -I$C $2 $C/conv_igemm.hip $C/conv_dma.hip $C/convt_quad.hip $C/conv_strip.hip $C/conv_bwd.hip $C/bn.hip $C/stft.hip $C/layout.hip $C/rl_ops.hip $C/rollout_fused.hip $C/pack_batch.hip $C/fftconv.hip $C/api.hip -o /tmp/libm2h_$1.so 2>/dev/null & }
build pipe1d2 "-DM2H_LAST32_PIPE=1 -DM2H_STRIP_DEPTH=2"
build pipe0d2 "-DM2H_LAST32_PIPE=0 -DM2H_STRIP_DEPTH=2"
build pipe1d4 "-DM2H_LAST32_PIPE=1 -DM2H_STRIP_DEPTH=4"
wait
for r in 1 2; do for v in pipe1d2 pipe0d2 pipe1d4; do echo "== $v"; M2H_LIB=/tmp/libm2h_$v.so python tools/strip_bench.py 2>&1 | grep -v amdgpu; done; done
