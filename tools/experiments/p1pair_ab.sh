# A/B of the phase-1 tile grouping of the step kernel (-DGCRNN_P1_GROUP=1|2|4; the library is built with 2): bash tools/p1pair_ab.sh   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/gated_gcrnns_amd/csrc
mkdir -p /tmp/p1ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_P1_GROUP=1 -o /tmp/p1ab/lib_base.so $C/*.hip $C/gcrnn_host.cpp &
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_P1_GROUP=2 -o /tmp/p1ab/lib_pair.so $C/*.hip $C/gcrnn_host.cpp &
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGCRNN_P1_GROUP=4 -o /tmp/p1ab/lib_quad.so $C/*.hip $C/gcrnn_host.cpp &
wait
for i in 1 2 3; do for v in base pair quad; do
  echo -n "$v: "; GCRNN_LIBPATH=/tmp/p1ab/lib_$v.so python3 $R/tools/step_kernel_probe.py 256 32 3 2>&1 | tail -1
done; done
GCRNN_LIBPATH=/tmp/p1ab/lib_quad.so python3 -m pytest $R/tests/test_fused.py -m gpu -x -q -k "step_matches or training_autograd or backward_data" 2>&1 | tail -2
