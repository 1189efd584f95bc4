"""tools/mfma_hazard_probe_gen.py -- writes tools/mfma_hazard_probe.hip: how many wait states v_mfma_f64_16x16x4_f64 needs on gfx950
before its result is read (VALU, a dependent MFMA's srcA/B, srcC), with and without independent MFMAs in between, and the
write-after-write / write-after-read windows.  hipcc's hazard recogniser inserts these for MFMAs it emits itself; the two-wave
leaf (csrc/gpx_leaf.h) issues its MFMAs from volatile asm to pin their order, so it has to provide them itself.  diagnostic."""
import os

def nops(k):
    out = []
    while k > 0:
        c = min(k, 16)
        out.append("s_nop %d" % (c - 1))
        k -= c
    return out

def dmov(reg, val):          # a double constant into v[reg:reg+1]
    import struct
    lo, hi = struct.unpack("<II", struct.pack("<d", val))
    return ["v_mov_b32 v%d, 0x%x" % (reg, lo), "v_mov_b32 v%d, 0x%x" % (reg + 1, hi)]

A, B, W, X, Z, OUT = 2, 4, 20, 40, 48, 30
SENT = 12345.0

def prologue():
    s = dmov(A, 1.2345678901234567) + dmov(B, 2.3456789012345678)
    for t in (W, X, Z):
        for r in range(0, 8, 2):
            s += dmov(t + r, SENT)
    return s + nops(32)

def mf(dst, a, b, c):
    cs = "0" if c is None else "v[%d:%d]" % (c, c + 7)
    return "v_mfma_f64_16x16x4_f64 v[%d:%d], v[%d:%d], v[%d:%d], %s" % (dst, dst + 7, a, a + 1, b, b + 1, cs)

def rd(src):
    return ["v_mov_b32 v%d, v%d" % (OUT, src), "v_mov_b32 v%d, v%d" % (OUT + 1, src + 1)]

tests = []   # (name, k, expected, asm lines)
for k in (40, 0, 4, 8, 12, 14, 15, 16, 17, 18, 19, 20):
    tests.append(("valu_read", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(k) + rd(W)))
for k in (40, 0, 1, 2, 3, 4, 5, 6, 8, 12, 18):
    tests.append(("valu_read_after_1_mfma", k, 8.0, prologue() + [mf(W, A, B, None), mf(X, A, B, None)] + nops(k) + rd(W)))
for k in (40, 0, 1, 2, 4):
    tests.append(("valu_read_after_2_mfma", k, 8.0, prologue() + [mf(W, A, B, None), mf(X, A, B, None), mf(Z, A, B, None)] + nops(k) + rd(W)))
for k in (40, 0, 4, 8, 12, 14, 15, 16, 17, 18, 19):
    tests.append(("mfma_srcB", k, 32.0, prologue() + [mf(W, A, B, None)] + nops(k) + [mf(X, A, W, None)] + nops(24) + rd(X)))
for k in (40, 0, 1, 2, 3, 4, 6, 8):
    tests.append(("mfma_srcB_after_1_mfma", k, 32.0, prologue() + [mf(W, A, B, None), mf(Z, A, B, None)] + nops(k) + [mf(X, A, W, None)] + nops(24) + rd(X)))
for k in (40, 0, 1, 2, 4, 8, 12, 16, 18):
    tests.append(("mfma_srcC_same_tuple", k, 16.0, prologue() + [mf(W, A, B, None)] + nops(k) + [mf(W, A, B, W)] + nops(24) + rd(W)))
for k in (40, 0, 4, 8, 12, 16, 17, 18, 19, 20):
    tests.append(("waw_valu_write_junk", k, 777.0, prologue() + [mf(W, A, B, None)] + nops(k) + dmov(W + 4, 777.12345678901234) + nops(24) + rd(W + 4)))
for k in (40, 0, 1, 2, 4):
    tests.append(("waw_after_1_mfma", k, 777.0, prologue() + [mf(W, A, B, None), mf(X, A, B, None)] + nops(k) + dmov(W + 4, 777.12345678901234) + nops(24) + rd(W + 4)))
for k in (40, 0, 1, 2, 4, 8):
    tests.append(("war_valu_overwrites_srcA", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(k) + dmov(A, 5.4321098765432109) + nops(24) + rd(W)))
# AGPR destination: v_accvgpr_read of the result, the result as srcB from the AGPR, accumulate in place in AGPRs
def mfa(dst, a, b, c, b_agpr=False):
    cs = "0" if c is None else "a[%d:%d]" % (c, c + 7)
    bs = ("a[%d:%d]" if b_agpr else "v[%d:%d]") % (b, b + 1)
    return "v_mfma_f64_16x16x4_f64 a[%d:%d], v[%d:%d], %s, %s" % (dst, dst + 7, a, a + 1, bs, cs)
def aprologue():
    s = []
    for t in (0, 8, 16):
        for r in range(8):
            s += ["v_accvgpr_write_b32 a%d, v%d" % (t + r, W + (r & 1))]
    return s + nops(32)
def ard(src):
    return ["v_accvgpr_read_b32 v%d, a%d" % (OUT, src), "v_accvgpr_read_b32 v%d, a%d" % (OUT + 1, src + 1)]
for k in (40, 0, 1, 2, 4, 8, 16, 18):
    tests.append(("agpr_read", k, 8.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(k) + ard(0)))
for k in (40, 0, 1, 2, 4):
    tests.append(("agpr_read_after_1_mfma", k, 8.0, prologue() + aprologue() + [mfa(0, A, B, None), mfa(8, A, B, None)] + nops(k) + ard(0)))
for k in (40, 0, 1, 2, 4, 8, 16, 18):
    tests.append(("agpr_srcB", k, 32.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(k) + [mfa(8, A, 0, None, True)] + nops(24) + ard(8)))
for k in (40, 0, 1, 2, 4, 8, 16):
    tests.append(("agpr_srcC_same_tuple", k, 16.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(k) + [mfa(0, A, B, 0)] + nops(24) + ard(0)))
# v_readlane of a result
for k in (40, 0, 1, 2, 4, 8, 16, 18):
    tests.append(("readlane_read", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(k) +
                  ["v_readlane_b32 s20, v%d, 5" % W, "v_readlane_b32 s21, v%d, 5" % (W + 1), "v_mov_b32 v%d, s20" % OUT, "v_mov_b32 v%d, s21" % (OUT + 1)]))
# LDS store of a result (ds_write reads the register): same question as the VALU read
for k in (40, 0, 8, 16, 17, 18, 19):
    tests.append(("ds_write_read", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(k) +
                  ["ds_write_b64 v60, v[%d:%d]" % (W, W + 1), "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[%d:%d], v60" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)"]))

# the copy an MFMA result leaves through: v_mov_b64 / two v_mov_b32 (of a settled value 8.0 in W after the waits), then the LDS store
for k in (40, 0, 1, 2, 3, 4, 6, 8):
    tests.append(("mov_b64_then_ds_write", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(24) + dmov(OUT, SENT) + nops(8) +
                  ["v_mov_b64 v[%d:%d], v[%d:%d]" % (OUT, OUT + 1, W, W + 1)] + nops(k) +
                  ["ds_write_b64 v60, v[%d:%d]" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[%d:%d], v60" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)"]))
for k in (40, 0, 1, 2, 4):
    tests.append(("mov_b32x2_then_ds_write", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(24) + dmov(OUT, SENT) + nops(8) +
                  ["v_mov_b32 v%d, v%d" % (OUT, W), "v_mov_b32 v%d, v%d" % (OUT + 1, W + 1)] + nops(k) +
                  ["ds_write_b64 v60, v[%d:%d]" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[%d:%d], v60" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)"]))
for k in (40, 0, 1, 2, 4):
    tests.append(("mfma_then_mov_b32x2_then_ds_write", k, 8.0, prologue() + [mf(W, A, B, None)] +
                  ["v_mov_b32 v%d, v%d" % (OUT, W), "v_mov_b32 v%d, v%d" % (OUT + 1, W + 1)] + nops(k) +
                  ["ds_write_b64 v60, v[%d:%d]" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[%d:%d], v60" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)"]))
for k in (40, 1, 2, 3, 4, 5, 6, 7):
    tests.append(("ds_write_read_fine", k, 8.0, prologue() + [mf(W, A, B, None)] + nops(k) +
                  ["ds_write_b64 v60, v[%d:%d]" % (W, W + 1), "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[%d:%d], v60" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)"]))

# a VALU write of an MFMA SOURCE right in front of the MFMA (what hipcc emits in front of an asm MFMA whose operand it has to
# copy into place): VGPR and AGPR sources, matrix pipe idle / busy with an independent MFMA
B3 = 6
def set_b3():
    return dmov(B3, 3.1415926535897931) + nops(8)
def setup_a01():   # a[0:1] = 2.0, a[8:9] = 3.0
    return ["v_accvgpr_write_b32 a0, v%d" % B, "v_accvgpr_write_b32 a1, v%d" % (B + 1), "v_accvgpr_write_b32 a8, v%d" % B3, "v_accvgpr_write_b32 a9, v%d" % (B3 + 1)] + nops(16)
for busy in (0, 1):
    for k in (40, 0, 1, 2, 3, 4, 6, 8):
        pre = [mf(X, A, B, None)] if busy else []
        tests.append(("valu_writes_vgpr_srcB%s" % ("_pipe_busy" if busy else ""), k, 12.0, prologue() + set_b3() + pre +
                      ["v_mov_b32 v%d, v%d" % (B, B3), "v_mov_b32 v%d, v%d" % (B + 1, B3 + 1)] + nops(k) + [mf(W, A, B, None)] + nops(24) + rd(W)))
    for k in (40, 0, 1, 2, 3, 4, 6, 8):
        pre = [mf(X, A, B, None)] if busy else []
        tests.append(("accvgpr_write_srcB%s" % ("_pipe_busy" if busy else ""), k, 12.0, prologue() + set_b3() + setup_a01() + pre +
                      ["v_accvgpr_write_b32 a0, v%d" % B3, "v_accvgpr_write_b32 a1, v%d" % (B3 + 1)] + nops(k) + [mfa(16, A, 0, None, True)] + nops(24) + ard(16)))
    for k in (40, 0, 1, 2, 3, 4, 6, 8):
        pre = [mf(X, A, B, None)] if busy else []
        tests.append(("accvgpr_mov_srcB%s" % ("_pipe_busy" if busy else ""), k, 12.0, prologue() + set_b3() + setup_a01() + pre +
                      ["v_accvgpr_mov_b32 a0, a8", "v_accvgpr_mov_b32 a1, a9"] + nops(k) + [mfa(16, A, 0, None, True)] + nops(24) + ard(16)))
    for k in (40, 0, 1, 2, 3, 4, 6, 8):
        pre = [mf(X, A, B, None)] if busy else []
        tests.append(("valu_writes_vgpr_srcA%s" % ("_pipe_busy" if busy else ""), k, 24.0, prologue() + set_b3() + pre +
                      ["v_mov_b32 v%d, v%d" % (A, B3), "v_mov_b32 v%d, v%d" % (A + 1, B3 + 1)] + nops(k) + [mf(W, A, B, None)] + nops(24) + rd(W)))

# the LAST register of the destination (rows 12 .. 15 of the tile, written by the last passes), accumulate form: C = 8 from a first
# MFMA, then C += 8; a reader that is let through too early sees 8 (or a partial sum), not 16
for k in (40, 0, 1, 2, 3, 4, 6, 8, 12, 16, 18, 20):
    tests.append(("valu_read_LAST_reg_acc", k, 16.0, prologue() + [mf(W, A, B, None)] + nops(32) + [mf(W, A, B, W)] + nops(k) + rd(W + 6)))
for k in (40, 0, 1, 2, 3, 4, 6, 8, 12, 16, 18, 20):
    tests.append(("agpr_read_LAST_reg_acc", k, 16.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + nops(k) + ard(6)))
for k in (40, 0, 1, 2, 3, 4, 6, 8, 12, 16, 18, 20):
    tests.append(("agpr_LAST_reg_as_srcB", k, 64.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + nops(k) + [mfa(8, A, 6, None, True)] + nops(24) + ard(8)))
for k in (40, 0, 1, 2, 3, 4, 6, 8, 12, 16, 18, 20):
    tests.append(("readlane_LAST_reg_acc", k, 16.0, prologue() + [mf(W, A, B, None)] + nops(32) + [mf(W, A, B, W)] + nops(k) +
                  ["v_readlane_b32 s20, v%d, 37" % (W + 6), "v_readlane_b32 s21, v%d, 37" % (W + 7), "v_mov_b32 v%d, s20" % OUT, "v_mov_b32 v%d, s21" % (OUT + 1)]))
for k in (40, 0, 1, 2, 3, 4, 6, 8, 12, 16, 18, 20):
    tests.append(("accvgpr_read_then_readlane_LAST_reg", k, 16.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + nops(k) +
                  ["v_accvgpr_read_b32 v%d, a6" % Z, "v_accvgpr_read_b32 v%d, a7" % (Z + 1), "v_readlane_b32 s20, v%d, 37" % Z, "v_readlane_b32 s21, v%d, 37" % (Z + 1),
                   "v_mov_b32 v%d, s20" % OUT, "v_mov_b32 v%d, s21" % (OUT + 1)]))

# the last register again, with one / two independent MFMAs issued behind the writer (each holds the pipe for 16 passes: is the
# requirement time or instruction count?), and registers 1 and 2
for nm in (1, 2):
    for k in (40, 0, 1, 2, 3, 4, 6, 8, 12):
        mid = [mf(X, A, B, None), mf(Z, A, B, None)][:nm]
        tests.append(("agpr_read_LAST_reg_after_%d_mfma" % nm, k, 0.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + mid + nops(k) + ard(6)))
for reg in (1, 2):
    for k in (40, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18):
        tests.append(("agpr_read_reg%d_acc" % reg, k, 0.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + nops(k) + ard(2 * reg)))
for k in (40, 0, 1, 2, 4, 8):
    tests.append(("back_to_back_accumulate_LAST_reg", k, 0.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + nops(k) + [mfa(0, A, B, 0)] + nops(32) + ard(6)))
for k in (40, 0, 4, 8, 12, 14, 16, 17, 18):
    tests.append(("ds_write_LAST_reg", k, 0.0, prologue() + [mf(W, A, B, None)] + nops(32) + [mf(W, A, B, W)] + nops(k) +
                  ["ds_write_b64 v60, v[%d:%d]" % (W + 6, W + 7), "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[%d:%d], v60" % (OUT, OUT + 1), "s_waitcnt lgkmcnt(0)"]))
for k in (40, 14, 15, 16, 17, 18):
    tests.append(("agpr_read_LAST_reg_fine", k, 0.0, prologue() + aprologue() + [mfa(0, A, B, None)] + nops(32) + [mfa(0, A, B, 0)] + nops(k) + ard(6)))

src = ['// generated by tools/mfma_hazard_probe_gen.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstring>',
       '__global__ void probe(double *out, unsigned *bad, const double *ref)', '{', '    __shared__ double sh[1024];', '    double r; sh[threadIdx.x] = 0.0; __syncthreads();', '    const int wave0 = (threadIdx.x & 63) == 0;']
clob = ", ".join(['"v%d"' % i for i in list(range(2, 8)) + list(range(20, 32)) + list(range(40, 56)) + [60]] + ['"a%d"' % i for i in range(24)] + ['"s20"', '"s21"'])
for i, (name, k, exp, lines) in enumerate(tests):
    body = "\\n\\t".join(["v_lshlrev_b32 v60, 3, %2"] + lines + ["v_mov_b32 %0, v" + str(OUT), "v_mov_b32 %1, v" + str(OUT + 1)])
    src.append('    { unsigned lo, hi; asm volatile("' + body + '" : "=v"(lo), "=v"(hi) : "v"(threadIdx.x) : ' + clob + ', "memory");')
    src.append('      r = __hiloint2double((int)hi, (int)lo); if (wave0) { if (!ref) out[%d] = r; else if (r != ref[%d]) atomicAdd(&bad[%d], 1u); } if (sh[5] == 7.0) out[0] = 1; }' % (i, i, i))
src.append('}')
src.append('struct T { const char *name; int k; double expect; };')
src.append('static const T tests[] = {')
for name, k, exp, _ in tests:
    src.append('    {"%s", %d, %.1f},' % (name, k, exp))
src.append('};')
src.append(r'''int main()
{
    const int n = sizeof(tests) / sizeof(tests[0]);
    double *d, *dref; unsigned *bad; hipMalloc(&d, n * sizeof(double)); hipMalloc(&dref, n * sizeof(double)); hipMalloc(&bad, n * sizeof(unsigned));
    static double h[1024], href[1024]; static unsigned hb[2][1024];
    // the reference of every group: the same sequence with 40 wait states at the place in question, one wave alone
    hipMemset(d, 0, n * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, bad, (const double *)nullptr);
    hipDeviceSynchronize(); hipMemcpy(h, d, n * sizeof(double), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) { int r = i; while (tests[r].k != 40 || strcmp(tests[r].name, tests[i].name)) --r; href[i] = h[r]; }
    hipMemcpy(dref, href, n * sizeof(double), hipMemcpyHostToDevice);
    // 1: one wave alone.  2: 1024 workgroups of 16 waves -- four waves a SIMD all issuing MFMAs (contended matrix pipe)
    for (int cfg = 0; cfg < 2; ++cfg) {
        hipMemset(bad, 0, n * sizeof(unsigned));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(cfg ? 1024 : 1), dim3(cfg ? 1024 : 64), 0, 0, d, bad, (const double *)dref);
        hipDeviceSynchronize(); hipMemcpy(hb[cfg], bad, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    }
    const char *last = "";
    for (int i = 0; i < n; ++i) {
        if (strcmp(last, tests[i].name)) { printf("\n%-28s reference %.17g (wait states: results that differ from it, of 3 x one wave alone / of 3 x 16384 contended waves)\n   ", tests[i].name, href[i]); last = tests[i].name; }
        if (tests[i].k != 40) printf(" %d:%u/%u", tests[i].k, hb[0][i], hb[1][i]);
    }
    printf("\n");
    return 0;
}''')
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_hazard_probe.hip"), "w").write("\n".join(src) + "\n")
print(len(tests), "tests")
