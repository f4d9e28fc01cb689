"""Writes tools/probes/ge_rank_pipe_timing.hip = ge_rank_pipe.hip with s_memtime stamps around its phases (one
workgroup reports setup / diagonal tile / tile / epilogue ticks through true_loss[0..6]) and builds
tools/probes/libge_timing.so from it plus the other objects in graphembeddings_amd/_obj.
tools/probes/rank_timing.py reads the stamps (GE_LIB=tools/probes/libge_timing.so).  Test infrastructure."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "graphembeddings_amd", "csrc")
s = open(os.path.join(CSRC, "ge_rank_pipe.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, 1)


rep("  while (idx < idx_end) {\n    const int rb",
    "  long long T0, T_setup = 0, T_diag = 0, T_tile = 0, T_epi = 0, T_epi2 = 0; int n_t = 0, n_seg = 0;\n"
    "  while (idx < idx_end) {\n    T0 = clock64(); ++n_seg;\n    const int rb")
rep("    __syncthreads();\n\n    auto cand_of = ",
    "    __syncthreads();\n    T_setup += clock64() - T0; T0 = clock64();\n\n    auto cand_of = ")
rep("    int raw_reg = 0;\n", "    int raw_reg = 0;\n    T_diag += clock64() - T0;\n")
rep("      const int64_t n0 = (int64_t)ct * kRB;\n      if constexpr (F16) f16_tile(row_of(cid_next), cid < 0 || cid >= N, max_norm, lds, R, acc);\n"
    "      else pipe_tile<CW, NCH>(table, N, d, lda, cid, max_norm, spec, lds, rA, rB, acc);\n",
    "      const int64_t n0 = (int64_t)ct * kRB;\n      T0 = clock64();\n"
    "      if constexpr (F16) f16_tile(row_of(cid_next), cid < 0 || cid >= N, max_norm, lds, R, acc);\n"
    "      else pipe_tile<CW, NCH>(table, N, d, lda, cid, max_norm, spec, lds, rA, rB, acc);\n"
    "      T_tile += clock64() - T0; T0 = clock64(); ++n_t;\n")
rep("      __syncthreads();\n      if (t < kRB) {\n        const unsigned* m = lds.bm + t * 4;",
    "      T_epi += clock64() - T0; T0 = clock64();\n      __syncthreads();\n      if (t < kRB) {\n"
    "        const unsigned* m = lds.bm + t * 4;")
rep("      // no barrier: the next tile's first write", "      T_epi2 += clock64() - T0;\n      // no barrier: the next tile's first write")
rep("    if (MODE != 2 && t < kRB && m0 + t < B) {\n      if (raw_reg) atomicAdd",
    "    if (true_loss && blockIdx.x == 7 && t == 0 && idx >= idx_end) {\n"
    "      true_loss[0] = (float)T_setup; true_loss[1] = (float)T_diag; true_loss[2] = (float)T_tile; true_loss[3] = (float)T_epi;\n"
    "      true_loss[4] = (float)n_t; true_loss[5] = (float)T_epi2; true_loss[6] = (float)n_seg;\n    }\n"
    "    if (MODE != 2 && t < kRB && m0 + t < B) {\n      if (raw_reg) atomicAdd")
# ablations (wrong results, timing only): -DGE_ABL=bits  2: no chunk barrier, 4: no LDS stores of
# the staged chunk, 8: no global requests
rep('#include "ge_rank_dev.h"\n', '#include "ge_rank_dev.h"\n#ifndef GE_ABL\n#define GE_ABL 0\n#endif\n')
rep("      if (!LAST && g == NG - 1) __syncthreads();", "      if (!LAST && g == NG - 1 && !(GE_ABL & 2)) __syncthreads();")
rep("          if (!LAST && fp < NV) rf[fp]", "          if (!LAST && fp < NV && !(GE_ABL & 8)) rf[fp]")
out = os.path.join(ROOT, "tools", "probes", "ge_rank_pipe_timing.hip")
open(out, "w").write(s)
flags = sys.argv[1:]
tag = "".join(c for c in "".join(flags) if c.isdigit())
obj = "/tmp/ge_rank_pipe_timing%s.o" % tag
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + CSRC, *flags, "-c", out, "-o", obj],
                      stderr=subprocess.DEVNULL)
objs = [o for o in glob.glob(os.path.join(ROOT, "graphembeddings_amd", "_obj", "*.o")) if not o.endswith("ge_rank_pipe.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                       os.path.join(ROOT, "tools", "probes", "libge_timing%s.so" % tag), *objs, obj])
print("built tools/probes/libge_timing%s.so" % tag)
