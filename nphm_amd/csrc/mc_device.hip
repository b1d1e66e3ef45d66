// mc_device.hip — marching cubes on the GPU for a DEVICE-resident SDF volume: the mesh half of
// mesh_from_logits (src/NPHM/utils/reconstruction.py:22-37 -> third-party PyMCubes) without the
// 64 MiB device->host copy of the volume and the host pass.  Same case table (mc_table.h, generated),
// same vertex / triangle ORDER and the same double-precision interpolation as the host extractor
// (marching_cubes.cpp): the two produce bit-identical meshes.
//
//   count  : per lattice point p: crossed +x/+y/+z edges (0..3), triangles of the cell based at p (0..5)
//   scan   : exclusive prefix sums (hipCUB DeviceScan) -> vertex / triangle offsets, totals
//   emit   : vertices (interpolated, float64, index space) + edge -> vertex map; triangles through the map
#define MC_TABLE_QUALIFIER __device__
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "capi_common.h"
#include "mc_table.h"

namespace nphm {
namespace mcdev {

struct Dims {
  int nx, ny, nz;
  float sgn;
  double iso;
};

__device__ __forceinline__ bool inside(const float* vol, const Dims& d, int64_t p) { return double(d.sgn * vol[p]) > d.iso; }

__global__ void count_kernel(const float* vol, Dims d, int* cnt_v, int* cnt_f) {
  const int64_t plane = int64_t(d.ny) * d.nz, total = plane * d.nx;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int i = int(p / plane), rem = int(p % plane), j = rem / d.nz, k = rem % d.nz;
  const bool hx = i + 1 < d.nx, hy = j + 1 < d.ny, hz = k + 1 < d.nz;
  const bool in0 = inside(vol, d, p);
  int cv = 0;
  if (hx && inside(vol, d, p + plane) != in0) ++cv;
  if (hy && inside(vol, d, p + d.nz) != in0) ++cv;
  if (hz && inside(vol, d, p + 1) != in0) ++cv;
  int cf = 0;
  if (hx && hy && hz) {
    unsigned m = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      m |= unsigned(inside(vol, d, p + (c & 1) * plane + ((c >> 1) & 1) * int64_t(d.nz) + ((c >> 2) & 1))) << c;
    cf = MC_NTRIS[m];
  }
  cnt_v[p] = cv;
  cnt_f[p] = cf;
}

__global__ void totals_kernel(const int* cnt_v, const int* off_v, const int* cnt_f, const int* off_f, int64_t total,
                              int64_t* out) {
  out[0] = int64_t(off_v[total - 1]) + cnt_v[total - 1];
  out[1] = int64_t(off_f[total - 1]) + cnt_f[total - 1];
}

__global__ void vertex_kernel(const float* vol, Dims d, const int* off_v, int* eid, double* verts) {
  const int64_t plane = int64_t(d.ny) * d.nz, total = plane * d.nx;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int i = int(p / plane), rem = int(p % plane), j = rem / d.nz, k = rem % d.nz;
  const int pos[3] = {i, j, k};
  const int dims[3] = {d.nx, d.ny, d.nz};
  const int64_t stride[3] = {plane, d.nz, 1};
  const double v0 = double(d.sgn * vol[p]);
  const bool in0 = v0 > d.iso;
  int id = off_v[p];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (pos[a] + 1 >= dims[a]) continue;
    const double v1 = double(d.sgn * vol[p + stride[a]]);
    if ((v1 > d.iso) == in0) continue;
    const double t = (d.iso - v0) / (v1 - v0);
    double* v = verts + int64_t(id) * 3;
    v[0] = i; v[1] = j; v[2] = k;
    v[a] += t;
    eid[p * 3 + a] = id++;
  }
}

__global__ void face_kernel(const float* vol, Dims d, const int* off_f, const int* eid, int64_t* faces) {
  const int64_t plane = int64_t(d.ny) * d.nz, total = plane * d.nx;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int i = int(p / plane), rem = int(p % plane), j = rem / d.nz, k = rem % d.nz;
  if (i + 1 >= d.nx || j + 1 >= d.ny || k + 1 >= d.nz) return;
  unsigned m = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    m |= unsigned(inside(vol, d, p + (c & 1) * plane + ((c >> 1) & 1) * int64_t(d.nz) + ((c >> 2) & 1))) << c;
  const int nt = MC_NTRIS[m];
  int64_t f = off_f[p];
  for (int t = 0; t < 3 * nt; ++t) {
    const int e = MC_TRIS[m][t];
    const int a = e >> 2, b = MC_EDGE_CORNERS[e][0];
    const int64_t q = p + (b & 1) * plane + ((b >> 1) & 1) * int64_t(d.nz) + ((b >> 2) & 1);
    faces[f * 3 + (t % 3)] = eid[q * 3 + a];
    if (t % 3 == 2) ++f;
  }
}

// workspace: [cnt_v N][off_v N][cnt_f N][off_f N][eid 3N] int32, [totals 2] int64, scan temp
struct Layout {
  size_t cnt_v, off_v, cnt_f, off_f, eid, totals, temp, temp_bytes, bytes;
};

static Layout layout(int64_t n) {
  Layout l;
  size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += (b + 255) / 256 * 256; return r; };
  l.cnt_v = take(n * 4); l.off_v = take(n * 4); l.cnt_f = take(n * 4); l.off_f = take(n * 4);
  l.eid = take(n * 12); l.totals = take(16);
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, static_cast<int*>(nullptr), static_cast<int*>(nullptr), int(n));
  l.temp_bytes = tb;
  l.temp = take(tb);
  l.bytes = o;
  return l;
}

}  // namespace mcdev
}  // namespace nphm

extern "C" {

size_t nphm_mc_device_workspace_bytes(int nx, int ny, int nz) {
  if (nx < 2 || ny < 2 || nz < 2) return 0;
  return nphm::mcdev::layout(int64_t(nx) * ny * nz).bytes;
}

int nphm_mc_device_count(const float* volume, int nx, int ny, int nz, double iso, int negate, void* workspace,
                         int64_t* n_verts, int64_t* n_faces, void* stream) {
  using namespace nphm::mcdev;
  if (!volume || !workspace || !n_verts || !n_faces) return nphm_fail_msg("nphm_mc_device_count: null pointer");
  if (nx < 2 || ny < 2 || nz < 2) return nphm_fail_msg("nphm_mc_device_count: volume must be at least 2^3");
  const int64_t n = int64_t(nx) * ny * nz;
  if (n > 0x7fffffffLL) return nphm_fail_msg("nphm_mc_device_count: volume too large");
  const Layout l = layout(n);
  char* ws = static_cast<char*>(workspace);
  int* cnt_v = reinterpret_cast<int*>(ws + l.cnt_v); int* off_v = reinterpret_cast<int*>(ws + l.off_v);
  int* cnt_f = reinterpret_cast<int*>(ws + l.cnt_f); int* off_f = reinterpret_cast<int*>(ws + l.off_f);
  int64_t* totals = reinterpret_cast<int64_t*>(ws + l.totals);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Dims d{nx, ny, nz, negate ? -1.f : 1.f, iso};
  const unsigned blocks = unsigned((n + 255) / 256);
  hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, st, volume, d, cnt_v, cnt_f);
  size_t tb = l.temp_bytes;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(ws + l.temp, tb, cnt_v, off_v, int(n), st);
  if (e != hipSuccess) return nphm_fail("nphm_mc_device_count scan", e);
  e = hipcub::DeviceScan::ExclusiveSum(ws + l.temp, tb, cnt_f, off_f, int(n), st);
  if (e != hipSuccess) return nphm_fail("nphm_mc_device_count scan", e);
  hipLaunchKernelGGL(totals_kernel, dim3(1), dim3(1), 0, st, cnt_v, off_v, cnt_f, off_f, n, totals);
  int64_t host[2];
  e = hipMemcpyAsync(host, totals, sizeof(host), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);           // the sizes decide the output allocation
  if (e != hipSuccess) return nphm_fail("nphm_mc_device_count", e);
  *n_verts = host[0];
  *n_faces = host[1];
  return 0;
}

int nphm_mc_device_emit(const float* volume, int nx, int ny, int nz, double iso, int negate, void* workspace,
                        double* verts, int64_t* faces, void* stream) {
  using namespace nphm::mcdev;
  if (!volume || !workspace) return nphm_fail_msg("nphm_mc_device_emit: null pointer");
  const int64_t n = int64_t(nx) * ny * nz;
  const Layout l = layout(n);
  char* ws = static_cast<char*>(workspace);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Dims d{nx, ny, nz, negate ? -1.f : 1.f, iso};
  const unsigned blocks = unsigned((n + 255) / 256);
  int* eid = reinterpret_cast<int*>(ws + l.eid);
  if (verts)
    hipLaunchKernelGGL(vertex_kernel, dim3(blocks), dim3(256), 0, st, volume, d, reinterpret_cast<const int*>(ws + l.off_v),
                       eid, verts);
  if (faces)
    hipLaunchKernelGGL(face_kernel, dim3(blocks), dim3(256), 0, st, volume, d, reinterpret_cast<const int*>(ws + l.off_f),
                       eid, faces);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nphm_fail("nphm_mc_device_emit launch", e);
  return 0;
}

}  // extern "C"
