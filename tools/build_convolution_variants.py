#!/usr/bin/env python3
"""Builds of the reference-shaped convolution example for tools/ab_convolution.py: examples/reference_shape_kernel.hip with its two
user_convolution_kernel* templates replaced by a variant, one build_ab/libconv_<variant>.so each (CPU only: hipcc cross-compiles).
    base       the filter fetched between the two transforms (rounds 3 ... 6 until the last day)
    hoist      fetched with the series, held in registers
    hoist_lb   ... and __launch_bounds__(256) on the kernels                       (= what examples/ ships)
    hlds       staged in a second shared array (contract form only)
    hlds_lb    ... and __launch_bounds__(256)
    w8         base with amdgpu_waves_per_eu(8, 8)
    hoist_w8   hoist with amdgpu_waves_per_eu(8, 8)
    v4         hoist_lb, __restrict__ pointers, 1/N folded into the filter
    v5         hoist_lb, 1/N folded into the filter
    v7         hoist_lb, __restrict__ pointers
Results: profiles/r06_convolution_user_kernel.txt."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build_ab")


def kernels(where="registers", attr="", restrict="", scale_in_filter=False):
    """the two kernels; where = 'between' | 'registers' | 'shared' (the contract form; the register form has no second array: 'shared' -> 'between')"""
    R = restrict
    scaled = "make_float2(g.x * (1.0f / N), g.y * (1.0f / N))" if scale_in_filter else "g"
    drain = ("        d_y[offset + threadIdx.x + k * Q] = s_data[threadIdx.x + k * Q];" if scale_in_filter else
             "        const float2 v = s_data[threadIdx.x + k * Q];\n        d_y[offset + threadIdx.x + k * Q] = make_float2(v.x * (1.0f / N), v.y * (1.0f / N));")
    product_r = ("x[k] = make_float2(a.x * h.x - a.y * h.y, a.x * h.y + a.y * h.x);" if scale_in_filter else
                 "x[k] = make_float2((a.x * h.x - a.y * h.y) * (1.0f / N), (a.x * h.y + a.y * h.x) * (1.0f / N));")
    if where == "registers":
        decl, fetch, use = "    float2 hh[4];\n", f"    for (int k = 0; k < 4; k++) {{ const float2 g = d_H[threadIdx.x + k * Q]; hh[k] = {scaled}; }}\n", "hh[k]"
    elif where == "shared":
        decl, fetch, use = "    __shared__ float2 s_H[N];\n", f"    for (int k = 0; k < 4; k++) {{ const float2 g = d_H[threadIdx.x + k * Q]; s_H[threadIdx.x + k * Q] = {scaled}; }}\n", "s_H[threadIdx.x + k * Q]"
    else:
        decl, fetch, use = "", "", "d_H[threadIdx.x + k * Q]"
    if where == "registers":
        decl_r, fetch_r, use_r = decl, fetch, use
    else:
        decl_r, fetch_r, use_r = "", "", "d_H[threadIdx.x + k * Q]"
        if scale_in_filter:
            raise SystemExit("1/N folded into the filter needs the filter in registers in the register form")
    return f"""template <class Fwd, class Inv>
__global__ void {attr}user_convolution_kernel(const float2* {R}d_x, const float2* {R}d_H, float2* {R}d_y) {{
    __shared__ float2 s_data[Fwd::fft_sm_required];
    constexpr int N = Fwd::fft_length, Q = Fwd::fft_length_quarter;
    const int offset = blockIdx.x * N;
{decl}    for (int k = 0; k < 4; k++) s_data[threadIdx.x + k * Q] = d_x[offset + threadIdx.x + k * Q];
{fetch}    __syncthreads();
    do_SMFFT_CT_DIT<Fwd>(s_data);
    __syncthreads();
    for (int k = 0; k < 4; k++) {{
        const float2 a = s_data[threadIdx.x + k * Q], h = {use};
        s_data[threadIdx.x + k * Q] = make_float2(a.x * h.x - a.y * h.y, a.x * h.y + a.y * h.x);
    }}
    __syncthreads();
    do_SMFFT_CT_DIT<Inv>(s_data);
    __syncthreads();
    for (int k = 0; k < 4; k++) {{
{drain}
    }}
}}
template <class Fwd, class Inv>
__global__ void {attr}user_convolution_kernel_registers(const float2* {R}d_x, const float2* {R}d_H, float2* {R}d_y) {{
    __shared__ float2 s_scratch[Fwd::fft_sm_required];
    constexpr int N = Fwd::fft_length, Q = Fwd::fft_length_quarter;
    const int offset = blockIdx.x * N;
    float2 x[4];
{decl_r}    for (int k = 0; k < 4; k++) x[k] = d_x[offset + threadIdx.x + k * Q];
{fetch_r}    do_SMFFT_CT_DIT_registers<Fwd>(x, s_scratch);
    for (int k = 0; k < 4; k++) {{
        const float2 a = x[k], h = {use_r};
        {product_r}
    }}
    __syncthreads();
    do_SMFFT_CT_DIT_registers<Inv>(x, s_scratch);
    for (int k = 0; k < 4; k++) d_y[offset + threadIdx.x + k * Q] = x[k];
}}
"""


LB, W8 = "__launch_bounds__(256) ", "__attribute__((amdgpu_waves_per_eu(8, 8))) "
VARIANTS = {
    "base": dict(where="between"),
    "hoist": dict(),
    "hoist_lb": dict(attr=LB),
    "hlds": dict(where="shared"),
    "hlds_lb": dict(where="shared", attr=LB),
    "w8": dict(where="between", attr=W8),
    "hoist_w8": dict(attr=W8),
    "v4": dict(attr=LB, restrict="__restrict__ ", scale_in_filter=True),
    "v5": dict(attr=LB, scale_in_filter=True),
    "v7": dict(attr=LB, restrict="__restrict__ "),
}


def main():
    names = sys.argv[1:] or list(VARIANTS)
    src = open(os.path.join(ROOT, "examples", "reference_shape_kernel.hip")).read()
    i0 = src.index("template <class Fwd, class Inv>\n__global__ void")
    i1 = src.index("// which = 0: shared-memory form, 1: register form")
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        path = os.path.join(OUT, f"conv_{name}.hip")
        open(path, "w").write(src[:i0] + kernels(**VARIANTS[name]) + src[i1:])
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
               "-shared", path, "-o", os.path.join(OUT, f"libconv_{name}.so"), "-Rpass-analysis=kernel-resource-usage"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-3000:])
            raise SystemExit(p.returncode)
        lines = p.stderr.splitlines()
        report = []
        for i, line in enumerate(lines):
            if "Function Name" in line and "user_convolution_kernel" in line:
                block = " ".join(lines[i:i + 12])
                import re
                v = re.search(r" VGPRs: (\d+)", block)
                s = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", block)
                report.append(("registers form" if "kernel_registers" in line else "contract form") + f": {v.group(1)} VGPRs, {s.group(1)} B scratch")
        print(f"{name}: " + "; ".join(report), flush=True)


if __name__ == "__main__":
    main()
