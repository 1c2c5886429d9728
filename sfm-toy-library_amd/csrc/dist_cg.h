// dist_cg.h -- the reduced-system CG of the sharded solve WITHOUT the redundant solve (SURVEY 8(e) "alternative to
// evaluate"): exchange (B) is a reduce-scatter of the upper-triangle blocks of S~ into contiguous ranges of block rows (every rank
// ends up owning the SUMMED blocks of its rows: 1/N of the matrix, half the bytes of the all-reduce), the matrix-vector product is
// formed from the owned blocks (row part + transposed part, no atomics) as a PARTIAL vector, and the one collective of a CG
// iteration is the all-reduce of that vector (d doubles: 48 KB at 1000 cameras).  Vector updates, the coarse space and the stopping
// test are replicated (same inputs, same arithmetic on every rank: identical decisions, no broadcast).
#pragma once
#include <hip/hip_runtime.h>
#include "device_arena.h"
#include "ba_kernels.h"
#include <vector>

namespace sfmba {

struct DistCg {
    int d = 0, ld = 0, ncam = 0, rank = 0, world = 1;
    int row0 = 0, row1 = 0;              // owned block rows (cameras)
    long long chunk_blocks = 0;          // blocks per rank in the reduce-scatter layout (the largest range, the others padded)
    std::vector<int> rows;               // [world + 1] partition of the block rows
    int* d_row_shift = nullptr;          // [ncam] block offset to ADD to a block's upper-triangle list position (pair pass -> reduce-scatter layout)
    double *qa = nullptr, *qb = nullptr; // [ld] row part / transposed part of the partial product
    double* qred = nullptr;              // [9 * ld] the all-reduce buffer: partial products (one per CG iteration; eight at the coarse setup)
    double *x = nullptr, *r = nullptr, *p = nullptr;   // [ld] each ([2][ld] for r: by launch parity), replicated (x: the caller's buffer -- DenseSolver::vec, where k_cam_update looks)
    double* AW = nullptr;                // [8][ld] S~ W~
    double* scal = nullptr;              // [128] rz, thresholds, E^-1 (64), ...
    double* state = nullptr;             // [2][16] {r.z, c = W~^T r} by launch parity (multi-workgroup form)
    int np = 0;                          // workgroups of the product kernels (four cameras each + one for the focal entry): rows of partial dots in the all-reduce
    int launched = 0;                    // CG launches of the running solve (a launch compares the done flag with its own number)
    bool ready = false;
};

// partition of the block rows that balances the number of off-diagonal upper blocks; chunk_blocks = the largest share
void dcg_partition(int ncam, int world, std::vector<int>* rows, long long* chunk_blocks);
int  dcg_create(DistCg* g, int d, int ld, int ncam, int rank, int world, DeviceArena* arena);
// values per rank of the reduce-scatter (36 per block)
inline long long dcg_chunk_values(const DistCg& g) { return 36ll * g.chunk_blocks; }

// owned: the rank's summed blocks (chunk `rank` of the reduce-scatter layout), fp64 or fp32.  F_focal: row d-1 of the CG's dense
// matrix (the glue of the pair pass left S~_fj there); bt: b~; W: the 8 gauge vectors or null.
typedef int (*dcg_allreduce_fn)(void* ctx, void* buf, long long n_doubles, hipStream_t s);
struct DcgSolveArgs {
    const void* owned = nullptr; bool owned_f32 = false;
    const double* focal_row = nullptr; const float* focal_row32 = nullptr;
    const double* bt = nullptr; const double* W = nullptr;
    int* flags = nullptr;               // DenseSolver::flags: [0] done, [1] iterations, [2] x buffer (always 0 here)
    int* info = nullptr;                // linear-solver status word (set on breakdown)
    double tol = 1e-8; int anchor = 0; double cap = 1.0;
    // non-null: the product is formed IMPLICITLY from the rank's own points (implicit_schur.hip) -- no block of S~
    // exists anywhere, `owned` is ignored and nothing of the reduced matrix was exchanged (options.shard_distributed_cg = 2)
    const ImplicitProduct* implicit = nullptr;
};
// setup of one solve (coarse space: 8 products + ONE all-reduce, then E^-1; x = 0, r = b~, z, p); returns 0 or an error of the collective
int  dcg_begin(hipStream_t s, DistCg* g, const DcgSolveArgs& a, dcg_allreduce_fn ar, void* ctx);
// n iterations: product from the owned blocks, all-reduce of the partial vector, replicated update.  Iterations behind the converging one
// return at once -- and still take part in their collective, so that every rank issues the same sequence.
int  dcg_iterate(hipStream_t s, DistCg* g, const DcgSolveArgs& a, int n, dcg_allreduce_fn ar, void* ctx);

}  // namespace sfmba
