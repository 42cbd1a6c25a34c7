#!/usr/bin/env python3
"""Build tools/bin/libwsmgmap_trace.so: the product library with wall-clock stamps (s_memrealtime) inserted at the
phase boundaries of gru_fwd_kernel / gru_bwd_kernel.  Diagnostic build only (tools/trace_rnn.py reads the stamps);
the shipped kernels contain no stamps.

    python tools/mk_rnn_trace.py && python tools/trace_rnn.py        # second step on the GPU box
"""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ws-mgmap_amd", "csrc")
import re, sys
s = open(os.path.join(CSRC, 'wsmg_rnn.hip')).read()
s=s.replace("namespace {\n\nconstexpr int H = 512;","__device__ unsigned long long* g_trace = nullptr;\nextern \"C\" int wsmg_debug_set_trace(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }\nnamespace {\n#define TRACE(slot) do { if (g_trace && blockIdx.x == 5 && tid == 0) g_trace[t * 8 + (slot)] = wall_clock64(); } while (0)\n\nconstexpr int H = 512;",1)
k=s.index("__global__ __launch_bounds__(256) void gru_fwd_kernel")
e=s.index("struct GruBwdArgs {")
b=s[k:e]
b=b.replace("    float (*hcur)[H] = hs[t & 1];\n","    float (*hcur)[H] = hs[t & 1];\n    TRACE(0);\n",1)
b=b.replace("      if (__syncthreads_or(good ? 0 : 1)) return;   // timeout or error elsewhere: every thread leaves\n    }\n","      if (__syncthreads_or(good ? 0 : 1)) return;   // timeout or error elsewhere: every thread leaves\n    }\n    TRACE(1);\n",1)
b=b.replace("    halve_row<32, 8>(acc, lane);","    TRACE(2);\n    halve_row<32, 8>(acc, lane);",1)
b=b.replace("    if (worker) {\n      const float ghr","    TRACE(3);\n    if (worker) {\n      const float ghr",1)
b=b.replace("    (void)mk;\n","    (void)mk;\n    TRACE(4);\n",1)
assert b.count("TRACE(") == 5, "gru_fwd_kernel changed: update the stamp anchors"
s=s[:k]+b+s[e:]
k=s.index("__global__ __launch_bounds__(256) void gru_bwd_kernel")
e=s.index("}  // namespace",k)
b=s[k:e]
b=b.replace("    float* xcur = a.xg + (size_t)t * NWG * XG_WG;\n","    float* xcur = a.xg + (size_t)t * NWG * XG_WG;\n    TRACE(0);\n",1)
b=b.replace("    if (!grid_barrier(a.sync, (unsigned)NWG * (unsigned)(a.T - t), tid, &ok_lds)) return;","    TRACE(1);\n    if (!grid_barrier(a.sync, (unsigned)NWG * (unsigned)(a.T - t), tid, &ok_lds)) return;\n    TRACE(2);",1)
b=b.replace("    const float s = reduce32(acc, lane);","    TRACE(3);\n    const float s = reduce32(acc, lane);",1)
b=b.replace("    carry = (dh_direct + s) * mk;\n","    carry = (dh_direct + s) * mk;\n    TRACE(4);\n",1)
assert b.count("TRACE(") == 5, "gru_bwd_kernel changed: update the stamp anchors"
s=s[:k]+b+s[e:]
os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
open(os.path.join(CSRC, 'build', 'wsmg_rnn_trace.hip'), 'w').write(s)
subprocess.check_call(['make', '-C', CSRC])
objs = [os.path.join(CSRC, 'build', f) for f in sorted(os.listdir(os.path.join(CSRC, 'build'))) if f.endswith('.o') and f not in ('wsmg_rnn.o', 'wsmg_rnn_trace.o')]
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
                       '-c', os.path.join(CSRC, 'build', 'wsmg_rnn_trace.hip'), '-o', os.path.join(CSRC, 'build', 'wsmg_rnn_trace.o')])
os.makedirs(os.path.join(ROOT, 'tools', 'bin'), exist_ok=True)
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(ROOT, 'tools', 'bin', 'libwsmgmap_trace.so')] + objs
                      + [os.path.join(CSRC, 'build', 'wsmg_rnn_trace.o')])
print('built tools/bin/libwsmgmap_trace.so')
