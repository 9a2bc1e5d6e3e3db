/* splash_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's subdomain-grid surface reconstruction
 * (splashsurf_lib::reconstruct_surface with SpatialDecomposition::UniformGrid, scalar / non-SIMD code
 * path), following the reference function by function.  Every function in splash_oracle.c cites the
 * reference file:line it restates.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the product (libsplashsurf_hip.so) never does.
 *
 * Parity pinning: validated in the build container against the reference's own pre-built wheel
 * (tools/gen_goldens.py; fixtures in tests/golden/): per-particle densities bit-identical, meshes
 * identical up to vertex order (see DESIGN.md "Oracle").
 */
#ifndef SPLASH_ORACLE_H
#define SPLASH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* f32 instantiation: so_*   (reference: reconstruct_surface::<i64, f32>) */
#define SO_REAL float
#define SOT(name) so_##name
#define SOFN(name) so_##name
#include "splash_oracle_decl.h"
#undef SO_REAL
#undef SOT
#undef SOFN
/* f64 instantiation: so64_* (reference: reconstruct_surface::<i64, f64>, always the scalar code path) */
#define SO_REAL double
#define SOT(name) so64_##name
#define SOFN(name) so64_##name
#include "splash_oracle_decl.h"
#undef SO_REAL
#undef SOT
#undef SOFN

#ifdef __cplusplus
}
#endif
#endif
