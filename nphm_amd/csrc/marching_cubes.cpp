// marching_cubes.cpp — host-side iso-surface extraction for mesh_from_logits
// (src/NPHM/utils/reconstruction.py:22-37 of the reference, which calls the third-party PyMCubes:
// `mcubes.marching_cubes(-logits, 0.0)`; PyMCubes is not vendored by the reference and absent here,
// so this is an independent marching-cubes with the same contract: vertices in index space
// (x, y, z) = (i, j, k), shared (indexed) vertices, triangles as vertex-index triples).
//
// Slab-parallel over the host cores, deterministic output (independent of the thread count):
//   pass A  inside bitmap (1 byte per lattice point)
//   pass B  per x-plane: lists of crossed lattice edges and of surface cells (8-point skips of uniform runs)
//   pass C  prefix sum over planes -> vertex ids, interpolated vertex positions
//   pass D  prefix sum -> triangles through the edge->vertex map (case table mc_table.h, generated)
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/nphm_amd.h"
#include "mc_table.h"

namespace {

struct McMesh {
  std::vector<double> verts;     // [nv][3]
  std::vector<int64_t> faces;    // [nf][3]
};

template <class F>
void parallel_planes(int n, int n_threads, F&& f) {
  if (n <= 0) return;
  n_threads = std::max(1, std::min(n_threads, n));
  if (n_threads == 1) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> pool;
  pool.reserve(n_threads);
  for (int t = 0; t < n_threads; ++t)
    pool.emplace_back([&]() {
      for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
    });
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int nphm_mc_extract(const float* volume, int nx, int ny, int nz, double iso, int negate, int n_threads,
                    void** handle, int64_t* n_verts, int64_t* n_faces) {
  if (!volume || !handle || !n_verts || !n_faces || nx < 2 || ny < 2 || nz < 2) return -2;
  // default: at most 16 threads — the passes are memory-bound and short (tens of ms at 256^3);
  // measured on a 256-core host: 35 ms with 1 thread, 18-21 ms with 8-64, 38 ms with 256
  if (n_threads <= 0) n_threads = std::min(16, int(std::thread::hardware_concurrency()));
  if (n_threads <= 0) n_threads = 1;
  const int64_t plane = int64_t(ny) * nz, total = plane * nx;
  const float sgn = negate ? -1.f : 1.f;
  auto val = [&](int64_t idx) -> double { return double(sgn * volume[idx]); };

  McMesh* mesh = new (std::nothrow) McMesh();
  if (!mesh) return -3;
  // in[p] = field(p) > iso, padded by 8 bytes so that rows can be scanned 8 points at a time;
  // eid[3p + a] = vertex id on the +a edge of lattice point p (written only where crossed)
  std::unique_ptr<uint8_t[]> in(new (std::nothrow) uint8_t[size_t(total) + 8]);
  std::unique_ptr<int32_t[]> eid(new (std::nothrow) int32_t[size_t(total) * 3]);
  if (!in || !eid) { delete mesh; return -3; }
  memset(in.get() + total, 0, 8);
  const int64_t stride[3] = {plane, nz, 1};
  const int dims[3] = {nx, ny, nz};

  // ---- pass A: inside bitmap -----------------------------------------------------------------------
  parallel_planes(nx, n_threads, [&](int i) {
    const float* v = volume + i * plane;
    uint8_t* o = in.get() + i * plane;
    if (negate) for (int64_t q = 0; q < plane; ++q) o[q] = double(-v[q]) > iso;
    else for (int64_t q = 0; q < plane; ++q) o[q] = double(v[q]) > iso;
  });

  // ---- pass B: per x-plane lists of crossed edges (3p + a) and of surface cells (p, mask) ----------
  struct PlaneWork { std::vector<int64_t> edges; std::vector<int64_t> cells; int64_t ntris = 0; };
  std::vector<PlaneWork> work(nx);
  auto load8 = [](const uint8_t* q) { uint64_t w; memcpy(&w, q, 8); return w; };
  parallel_planes(nx, n_threads, [&](int i) {
    PlaneWork& w = work[i];
    const bool has_x = i + 1 < nx;
    for (int j = 0; j < ny; ++j) {
      const bool has_y = j + 1 < ny;
      const uint8_t* r00 = in.get() + i * plane + int64_t(j) * nz;
      const uint8_t* r01 = has_y ? r00 + nz : r00;
      const uint8_t* r10 = has_x ? r00 + plane : r00;
      const uint8_t* r11 = r10 + (has_y ? nz : 0);
      for (int k = 0; k < nz;) {
        // 8 points at a time: skip runs in which the four rows are uniformly inside / outside
        // (the byte after the run takes part through the next iteration's first point)
        if (k + 8 < nz) {
          const uint64_t a = load8(r00 + k), b = load8(r01 + k), c = load8(r10 + k), d = load8(r11 + k);
          const uint64_t all1 = 0x0101010101010101ull;
          if (((a | b | c | d) == 0 || (a & b & c & d) == all1) && r00[k + 8] == r00[k] && r01[k + 8] == r00[k] &&
              r10[k + 8] == r00[k] && r11[k + 8] == r00[k]) {
            k += 8;
            continue;
          }
        }
        const int kend = std::min(nz, k + 8);
        for (; k < kend; ++k) {
          const int64_t p = i * plane + int64_t(j) * nz + k;
          const uint8_t in0 = r00[k];
          const bool has_z = k + 1 < nz;
          if (has_x && r10[k] != in0) w.edges.push_back(3 * p);
          if (has_y && r01[k] != in0) w.edges.push_back(3 * p + 1);
          if (has_z && r00[k + 1] != in0) w.edges.push_back(3 * p + 2);
          if (has_x && has_y && has_z) {
            // corner c at (c&1, (c>>1)&1, (c>>2)&1) = (x, y, z) offsets
            const unsigned m = unsigned(r00[k]) | unsigned(r10[k]) << 1 | unsigned(r01[k]) << 2 |
                               unsigned(r11[k]) << 3 | unsigned(r00[k + 1]) << 4 | unsigned(r10[k + 1]) << 5 |
                               unsigned(r01[k + 1]) << 6 | unsigned(r11[k + 1]) << 7;
            if (m != 0 && m != 255) {
              w.cells.push_back(p << 8 | m);
              w.ntris += MC_NTRIS[m];
            }
          }
        }
      }
    }
  });

  // ---- prefix sums -----------------------------------------------------------------------------------
  std::vector<int64_t> voff(nx + 1, 0), foff(nx + 1, 0);
  for (int i = 0; i < nx; ++i) {
    voff[i + 1] = voff[i] + int64_t(work[i].edges.size());
    foff[i + 1] = foff[i] + work[i].ntris;
  }
  const int64_t nv = voff[nx], nf = foff[nx];
  if (nv > 0x7fffffffLL) { delete mesh; return -4; }
  mesh->verts.resize(size_t(nv) * 3);
  mesh->faces.resize(size_t(nf) * 3);

  // ---- pass C: vertex ids + interpolated positions ---------------------------------------------------
  parallel_planes(nx, n_threads, [&](int i) {
    int64_t id = voff[i];
    for (const int64_t e : work[i].edges) {
      const int64_t p = e / 3;
      const int a = int(e % 3);
      const double v0 = val(p), v1 = val(p + stride[a]);
      const double t = (iso - v0) / (v1 - v0);
      const int64_t rem = p - i * plane;
      double* v = &mesh->verts[size_t(id) * 3];
      v[0] = i; v[1] = double(rem / nz); v[2] = double(rem % nz);
      v[a] += t;
      eid[size_t(e)] = int32_t(id++);
    }
  });
  (void)dims;

  // ---- pass D: triangles -------------------------------------------------------------------------------
  parallel_planes(nx, n_threads, [&](int i) {
    int64_t f = foff[i];
    for (const int64_t cm : work[i].cells) {
      const int64_t p = cm >> 8;
      const unsigned m = unsigned(cm & 255);
      const int nt = MC_NTRIS[m];
      for (int t = 0; t < 3 * nt; ++t) {
        const int e = MC_TRIS[m][t];
        const int a = e >> 2;                        // axis of the edge
        const int b = MC_EDGE_CORNERS[e][0];         // base corner
        const int64_t q = p + (b & 1) * plane + ((b >> 1) & 1) * int64_t(nz) + ((b >> 2) & 1);
        mesh->faces[size_t(f) * 3 + (t % 3)] = eid[size_t(q) * 3 + a];
        if (t % 3 == 2) ++f;
      }
    }
  });

  *handle = mesh;
  *n_verts = nv;
  *n_faces = nf;
  return 0;
}

int nphm_mc_fetch(void* handle, double* verts, int64_t* faces) {
  if (!handle) return -2;
  McMesh* mesh = static_cast<McMesh*>(handle);
  if (verts && !mesh->verts.empty()) memcpy(verts, mesh->verts.data(), mesh->verts.size() * sizeof(double));
  if (faces && !mesh->faces.empty()) memcpy(faces, mesh->faces.data(), mesh->faces.size() * sizeof(int64_t));
  return 0;
}

void nphm_mc_free(void* handle) { delete static_cast<McMesh*>(handle); }

}  // extern "C"
