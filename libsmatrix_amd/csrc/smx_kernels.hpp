// smx_kernels.hpp -- gfx950 device code of the (x,y)->uint32 path.
//
// HBM layout (DESIGN.md "Data layout"):
//   directory  : power-of-two array of 16-byte DirSlot {meta, x, base, used};
//                hash = murmur3 finaliser of x, linear probing, load <= 1/2.
//                Replaces the reference's x-directory smatrix_cmap_t
//                (src/smatrix.h:51-65, src/smatrix.c:598-741) -- its layout is
//                not observable through the API, so it is free to differ.
//   row tables : one block of 16*2^k 8-byte {key,value} cells per row in one
//                contiguous arena, addressed in 128-byte units.  A row table is
//                BIT-COMPATIBLE with the reference's smatrix_rmap_t data
//                (src/smatrix.h:35-49): identity hash `y % size`, linear
//                probing, empty == (0,0), growth x2 when `used > size/2` is
//                seen by an insert (src/smatrix.c:343-416).  Keeping it makes
//                rowlen/getrow order/file blocks identical to the reference.
//
// No locks: the reference's per-row spin RW lock (src/smatrix.c:843-889) is
// replaced by 64-bit CAS slot claims + 32-bit atomics on the value word, and
// structure changes (row creation, growth, directory growth) run in their own
// launches between rounds of the op kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace smx {

// The device code, by concern (each fragment says what it holds):
#include "kernels/layout.hpp"
#include "kernels/ops.hpp"
#include "kernels/prep_bulk.hpp"
#include "kernels/growth.hpp"
#include "kernels/rows.hpp"
#include "kernels/io_router.hpp"
#include "kernels/probes.hpp"

}  // namespace smx
