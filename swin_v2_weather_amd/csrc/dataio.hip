// ERA5 input assembly on the GPU: what the reference does on the host per sample (utils/data_loader_era5.py:163-171 crop +
// channel select, :98-107 z-score) or with DALI's fn.normalize (utils/data_loader_era5_dali.py:77-90), and what
// PreProcessor.forward concatenates afterwards (utils/preprocess_utils.py:50-68: input cos-zenith channel, one-hot land mask,
// standardised orography), as ONE pass from the raw staged fields to the model's input / target buffers.
//
// The raw fields arrive by async H2D copies of whole time slabs [Craw][Hraw][Wraw] fp32 (721 x 1440 rows, uncropped: the
// crop to img_size is an index bound here, not a host-side copy).  HBM-bound: reads 4 B, writes 4 B per element, 16-byte
// accesses along the longitude axis.
#include "common.h"

namespace {

// out[b][coff + s*Csel + c][i][j] = (raw[b][s][chan[c]][i][j] - mean[c]) / std[c]      (i < H, j < W)
__global__ __launch_bounds__(256) void era5_select_normalize_kernel(
    const float* __restrict__ raw, float* __restrict__ out, const int* __restrict__ chan, const float* __restrict__ mean,
    const float* __restrict__ stdv, int S, int Csel, int Craw, int Hraw, int Wraw, int H, int W, int Cout_total, int coff) {
    const int plane = blockIdx.y;                          // (b, s, c)
    const int c = plane % Csel, s = (plane / Csel) % S, b = plane / (Csel * S);
    const float m = mean[c], sd = stdv[c];
    const float* src = raw + (((size_t)b * S + s) * Craw + chan[c]) * (size_t)Hraw * Wraw;
    float* dst = out + ((size_t)b * Cout_total + coff + s * Csel + c) * (size_t)H * W;
    const int W4 = W >> 2;                                 // W % 4 == 0 and Wraw % 4 == 0 (checked on the host)
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * W4; idx += gridDim.x * blockDim.x) {
        const int i = idx / W4, j4 = idx - i * W4;
        f32x4 v = *(const f32x4*)(src + (size_t)i * Wraw + 4 * j4);
        // the reference's op order (img -= means; img /= stds) with IEEE division: bit-identical to the numpy path
        v[0] = (v[0] - m) / sd; v[1] = (v[1] - m) / sd; v[2] = (v[2] - m) / sd; v[3] = (v[3] - m) / sd;
        *(f32x4*)(dst + (size_t)i * W + 4 * j4) = v;
    }
}

// cos of the solar zenith angle on the 0.25 degree grid (lat 90 .. -90, lon 0 .. 359.75): the per-pixel half of the
// reference's zenith channel (data_loader_era5.py:109-146 -> modulus cos_zenith_angle):
//     out[b][coff + k][i][j] = sin(lat_i) sin(dec) + cos(lat_i) cos(dec) cos(ha0 + lon_j)
// sun[(b*nz + k)*3 + {0,1,2}] = sin(dec), cos(dec), hour angle at longitude 0 (reduced to (-pi, pi]) of time point k of
// sample b: the solar position itself is a handful of float64 scalar operations per time point and stays on the host
// (utils/data_loader_era5.py::sun_position), where the 1e5-hour arguments keep their precision.
__global__ __launch_bounds__(256) void era5_zenith_kernel(float* __restrict__ out, const float* __restrict__ sun, int nz,
                                                          int H, int W, int Cout_total, int coff) {
    const int plane = blockIdx.y, k = plane % nz, b = plane / nz;
    const float sd = sun[plane * 3], cd = sun[plane * 3 + 1], ha0 = sun[plane * 3 + 2];
    float* dst = out + ((size_t)b * Cout_total + coff + k) * (size_t)H * W;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * W; idx += gridDim.x * blockDim.x) {
        const int i = idx / W, j = idx - i * W;
        const float lat = (90.0f - 0.25f * i) * 0.017453292519943295f, lon = (0.25f * j) * 0.017453292519943295f;
        dst[idx] = sinf(lat) * sd + cosf(lat) * cd * cosf(ha0 + lon);
    }
}

// out[b][coff + c][:, :] = stat[c][:H, :W]  (static features, broadcast over the batch; preprocess_utils.py:62-63)
__global__ __launch_bounds__(256) void era5_static_kernel(const float* __restrict__ stat, float* __restrict__ out, int Cs,
                                                          int H, int W, int Cout_total, int coff) {
    const int plane = blockIdx.y, c = plane % Cs, b = plane / Cs;
    const float* src = stat + (size_t)c * H * W;
    float* dst = out + ((size_t)b * Cout_total + coff + c) * (size_t)H * W;
    const int n4 = (H * W) >> 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += gridDim.x * blockDim.x)
        ((f32x4*)dst)[idx] = ((const f32x4*)src)[idx];
}

}  // namespace

extern "C" int swv2_era5_select_normalize(const float* raw, float* out, const int* chan, const float* mean, const float* stdv, int B,
                                          int S, int Csel, int Craw, int Hraw, int Wraw, int H, int W, int Cout_total, int coff,
                                          void* stream) {
    SWV2_CHECK_ARG(raw && out && chan && mean && stdv, "swv2_era5_select_normalize: null pointer");
    SWV2_CHECK_ARG(B > 0 && S > 0 && Csel > 0 && H > 0 && H <= Hraw && W > 0 && W <= Wraw, "swv2_era5_select_normalize: bad sizes");
    SWV2_CHECK_ARG(W % 4 == 0 && Wraw % 4 == 0, "swv2_era5_select_normalize: W=%d, Wraw=%d must be multiples of 4", W, Wraw);
    SWV2_CHECK_ARG(coff >= 0 && coff + S * Csel <= Cout_total, "swv2_era5_select_normalize: channel range outside the output");
    dim3 grid(cdiv((long)H * (W / 4), 256 * 4), B * S * Csel);
    hipLaunchKernelGGL(era5_select_normalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, raw, out, chan, mean, stdv, S, Csel,
                       Craw, Hraw, Wraw, H, W, Cout_total, coff);
    SWV2_CHECK_LAUNCH("swv2_era5_select_normalize");
    return SWV2_OK;
}

extern "C" int swv2_era5_zenith(float* out, const float* sun, int B, int nz, int H, int W, int Cout_total, int coff, void* stream) {
    SWV2_CHECK_ARG(out && sun && B > 0 && nz > 0 && H > 0 && W > 0, "swv2_era5_zenith: bad argument");
    SWV2_CHECK_ARG(coff >= 0 && coff + nz <= Cout_total, "swv2_era5_zenith: channel range outside the output");
    dim3 grid(cdiv((long)H * W, 256 * 8), B * nz);
    hipLaunchKernelGGL(era5_zenith_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, sun, nz, H, W, Cout_total, coff);
    SWV2_CHECK_LAUNCH("swv2_era5_zenith");
    return SWV2_OK;
}

extern "C" int swv2_era5_static(const float* stat, float* out, int B, int Cs, int H, int W, int Cout_total, int coff, void* stream) {
    SWV2_CHECK_ARG(stat && out && B > 0 && Cs > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "swv2_era5_static: bad argument");
    SWV2_CHECK_ARG(coff >= 0 && coff + Cs <= Cout_total, "swv2_era5_static: channel range outside the output");
    dim3 grid(cdiv((long)H * W / 4, 256 * 4), B * Cs);
    hipLaunchKernelGGL(era5_static_kernel, grid, dim3(256), 0, (hipStream_t)stream, stat, out, Cs, H, W, Cout_total, coff);
    SWV2_CHECK_LAUNCH("swv2_era5_static");
    return SWV2_OK;
}
