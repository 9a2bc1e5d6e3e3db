// ss_global.h -- the reference's GLOBAL (non-decomposed) strategy on the GPU:
// reconstruct_surface_global (reconstruction.rs:65-194) as executed with enable_multi_threading = false,
// i.e. the reference's sequential, deterministic functions.  Used for SpatialDecomposition::None and for
// small domains when `auto_disable` is set (lib.rs:419-462).  Kernels in ss_global.hip, host flow in ss_api.hip.
//
// The hash maps of the reference (cell -> particles, grid point -> value, cell -> CellData) become dense
// arrays over the search grid / the marching-cubes grid.  Missing map entries and zeros are interchangeable
// in every test the reference performs on them (narrow_band_extraction.rs:79-88, 161-176).
#pragma once

#include "ss_device.h"

#define SS_GTILE 8          // level-set tile edge in grid points (one 512-thread workgroup per tile)
#define SS_GCHUNK 256       // particles examined per pass of the tile kernel (ascending index order)

template <class R>
struct SSGlobT {
    // marching-cubes grid = grid_for_reconstruction (lib.rs:476-516), NOT padded to subdomains
    R gmin[3];
    R cs;
    int np[3];
    int nc[3];
    // neighbourhood-search grid: UniformGrid::from_aabb(grid.aabb(), h) (neighborhood_search.rs:173-176)
    R smin[3];
    int snc[3];
    // kernel / level-set constants
    R h, h2, sigma, w0, mass, threshold;
    // SparseDensityMapGenerator (density_map.rs:582-640)
    R amin[3], amax[3];  // allowed domain: grid.aabb() shrunk by the kernel evaluation radius
    R radius_sq;         // (cs * ceil(h/cs) * (1 + sqrt(eps)))^2
    int half_cells;      // ceil(h/cs)
    int supported;       // points per dim touched by one particle: 2*half_cells + 2
    uint32_t n;          // particles (after the AABB filter)
};

template <class R>
void ssg_launch_cell_keys(const SSGlobT<R>& P, const R* xyz, uint32_t* keys, uint32_t* vals, uint32_t* cell_count, uint32_t* err, hipStream_t st);
// mode 0: densities + neighbour counts; mode 2: neighbour indices into CSR rows nb_ptr
template <class R>
void ssg_launch_density(const SSGlobT<R>& P, const R* xyz, const uint32_t* cell_start, const uint32_t* cell_items, R* rho, int mode, uint32_t* nb_count,
                        const unsigned long long* nb_ptr, uint32_t* nb_idx, hipStream_t st);
// per chunk of SS_GCHUNK consecutive particles: union of the stencil boxes of its contributing particles (6 ints, lo/hi exclusive)
template <class R>
void ssg_launch_chunk_boxes(const SSGlobT<R>& P, const R* xyz, int* boxes, hipStream_t st);
template <class R>
void ssg_launch_levelset(const SSGlobT<R>& P, const R* xyz, const R* rho, const int* boxes, R* G, hipStream_t st);
template <class R>
void ssg_launch_edge_masks(const SSGlobT<R>& P, const R* G, uint8_t* emask, uint32_t* vcount, hipStream_t st);
template <class R>
void ssg_launch_cell_count(const SSGlobT<R>& P, const R* G, const uint8_t* emask, uint32_t* tcount, uint32_t* err, hipStream_t st);
template <class R>
void ssg_launch_emit_vertices(const SSGlobT<R>& P, const R* G, const uint8_t* emask, const uint32_t* vbase, R* vertices, unsigned long long* vkeys, hipStream_t st);
template <class R>
void ssg_launch_emit_triangles(const SSGlobT<R>& P, const R* G, const uint8_t* emask, const uint32_t* vbase, const uint32_t* tcount, const uint32_t* tbase,
                               uint32_t* triangles, hipStream_t st);
