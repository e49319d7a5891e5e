/*
 * gdhost.h -- C ABI of libgdhost.so: the HOST side of the MI355X
 * marginalized-graph-kernel path that the reference does in Python per graph
 * and per call -- packing graphs into the device format, numbering label
 * classes, assigning every job a solver variant and laying the jobs out in
 * launch order.  Plain C, no HIP, no Python: pointers and sizes only, bound
 * with ctypes by graphdot_amd/hip/hostlib.py.
 *
 * What each entry point replaces in the reference (yhtang/GraphDot v0.8.1):
 *   gdh_pack_graphs      OctileGraph.__init__, once per graph
 *                        (graphdot/kernel/marginalized/_octilegraph.py:37-177)
 *   gdh_number_records   -- (new: label classes of the table kernels)
 *   gdh_classify_oc      the per-call launch configuration
 *   gdh_pairwise_jobs    the job list of a matrix evaluation (_kernel.py:172-182)
 *   gdh_pair_keys        (graphdot/kernel/marginalized/_backend_cuda.py:292-316:
 *   gdh_order_jobs        block / shared-memory sizing; the reference has one
 *                         solver and a global atomic job counter, this build
 *                         a menu of register-resident variants and a static,
 *                         cost-ordered job list per variant)
 *
 * Every function returns 0 on success, a negative code on bad arguments
 * (-1), a capacity that is too small (-2), or an exception inside the library
 * (-3: out of memory, no worker thread to be had and the like -- nothing is
 * thrown across this boundary).  Worker threads that cannot be created cost
 * nothing but time: their ranges run on the calling thread.  The numpy implementations in
 * graphdot_amd/kernel/marginalized/_devicegraph.py and _backend_hip.py stay
 * as the specification: the tests hold the native results to them byte for
 * byte.
 */
#ifndef GDHOST_H_
#define GDHOST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *gdh_version(void);

/* Pack G graphs into device images (csrc/device/graph.h).
 *
 * Inputs, concatenated over the graphs: node_off / edge_off [G + 1] prefix
 * sums of node / edge counts; node_id [Nn] the '!i' column (local node id of
 * every node row); ei, ej [Ne] local endpoints of every edge row; w [Ne] edge
 * weights or NULL (unweighted: 1); node_rec [Nn * node_size] the node records
 * (node_t, AoS) in INPUT row order; label_rec [Ne * label_size] the edge label
 * records in input row order.  An edge record is edge_size bytes: the weight
 * (weight_bytes = 0 none, 4 float, 8 double) at offset 0 and the label at
 * label_offset.
 *
 * Per graph: degree = float32 sum of incident weights (self loop once,
 * 0 -> 1); both orientations of every edge, duplicates collapse onto their
 * first occurrence; nodes renumbered by descending adjacency count (stable);
 * CSR of the directed nonzeros in the new numbering; sections
 *   degree f32[n] | node_t[n] | rowptr u16[n+1] | nz u16x2[nnz] | edge_t[nnz]
 *   | perm u16[n]
 * each 16-byte aligned, blobs back to back in `blob` (zero padded).
 *
 * Outputs: blob (capacity blob_capacity bytes), blob_off [G + 1], sec_off
 * [G * 6] (section offsets inside the graph's blob), nnz [G], and flat
 * per-node / per-nonzero arrays in the NEW numbering: perm u16 [Nn] (new ->
 * old), rank i64 [Nn] (old -> new), degree f32 [Nn], count i64 [Nn]
 * (adjacency counts), rowptr u16 [Nn + G], nz u16 [2 * nz_capacity] (i, j
 * pairs), eid i64 [nz_capacity] (local edge row of every nonzero), nz_off
 * [G + 1], maxdeg i64 [G]. */
int gdh_pack_graphs(int32_t G, const int64_t *node_off, const int64_t *edge_off,
                    const int64_t *node_id, const int64_t *ei, const int64_t *ej,
                    const float *w, const uint8_t *node_rec, int32_t node_size,
                    const uint8_t *label_rec, int32_t label_size,
                    int32_t edge_size, int32_t label_offset, int32_t weight_bytes,
                    uint8_t *blob, int64_t blob_capacity, int64_t *blob_off,
                    int64_t *sec_off, int64_t *nnz, uint16_t *perm, int64_t *rank,
                    float *degree, int64_t *count, uint16_t *rowptr, uint16_t *nz,
                    int64_t *eid, int64_t nz_capacity, int64_t *nz_off,
                    int64_t *maxdeg);

/* Number the distinct keys of n records.  The key of a record is the
 * concatenation of n_parts byte ranges (part_off / part_len inside the
 * record, itemsize bytes apart).  Classes are numbered in the order numpy's
 * np.unique gives them: keys of at most 8 bytes as little-endian unsigned
 * integers, longer keys lexicographically by byte.  cls [n] receives the class
 * of every record, first [n] (capacity) the index of the first record of every
 * class; *n_classes their number. */
int gdh_number_records(const uint8_t *rec, int64_t n, int32_t itemsize,
                       const int32_t *part_off, const int32_t *part_len,
                       int32_t n_parts, int32_t *cls, int64_t *first,
                       int64_t *n_classes);

/* Assign every pair of graph classes (ca[t], cb[t]) the first owner-computes
 * solver variant of the menu it fits (mgk_oc.h), or -1.
 * Per graph class c: n_node, n_nz, image bytes (incl. the label-class section
 * when the table kernels are used), largest degree, degree histogram hist
 * [16 per class] (hist[15]: 15 and above).
 * Per variant v: waves W, slots S, rows R, degree bound D, static layout
 * L [12 per variant, zero padded] of n_L entries (0: dynamic layout).  S = 0
 * marks an on-the-fly variant: the pairs with a degree above fly_min_degree
 * (8: where the slot variants end).
 * C: right-hand sides (1 value, 2 value + gradient); real_size 4 or 8;
 * lds_limit bytes per workgroup; extra_lds [n_var] (or NULL): LDS bytes a
 * variant needs beyond p, row sums, row map and images (slot values kept in
 * LDS, mgk_oc.h SL).
 * Outputs per pair: choice (variant index or -1), NP (rows with the odd LDS
 * stride, n1 * (n2 | 1)). */
int gdh_classify_oc(int64_t n_pairs, const int32_t *ca, const int32_t *cb,
                    const int32_t *n_node, const int32_t *n_nz,
                    const int64_t *image_bytes, const int32_t *maxdeg,
                    const uint16_t *hist, int32_t n_var, const int32_t *W,
                    const int32_t *S, const int32_t *R, const int32_t *D,
                    const int32_t *n_L, const int32_t *L, int32_t C,
                    int32_t real_size, int64_t lds_limit, int32_t fly_min_degree,
                    const int64_t *extra_lds, int32_t *choice, int64_t *NP);

/* Class-pair key of every job: pk[t] = cid[jobs[t].i] * nc + cid[jobs[t].j]
 * (jobs: n_jobs (u32 i, u32 j) pairs), and count [nc * nc] the number of jobs
 * per key. */
int gdh_pair_keys(const uint32_t *jobs, int64_t n_jobs, const int32_t *cid,
                  int32_t n_graphs, int32_t nc, int32_t *pk, int64_t *count);

/* Launch order of the jobs: a stable counting sort of the job ids by
 * rank_of_key[pk[t]] (ranks 0 .. n_ranks - 1: solver variant, then descending
 * cost; jobs of one rank keep their order).  order [n_jobs] receives the job
 * ids; with jobs / jobs_sorted (both or neither NULL) the (i, j) records are
 * written in that order too -- the list the solver kernels read. */
int gdh_order_jobs(const int32_t *pk, int64_t n_jobs, const int32_t *rank_of_key,
                   int64_t n_keys, int64_t n_ranks, uint32_t *order,
                   const uint32_t *jobs, uint32_t *jobs_sorted);

/* One section (col: 0 degree, 1 node, 2 rowptr, 3 nz, 4 edge, 5 perm) of every
 * graph of a packed batch back to back: count[g] records of itemsize bytes
 * from blob + blob_off[g] + sec_off[6 g + col].  (The label-class numbering
 * reads all node / edge records of a call as one array; numpy concatenated
 * a thousand slices for it.) */
int gdh_gather_section(const uint8_t *blob, const int64_t *blob_off, const int64_t *sec_off,
                       int32_t col, const int64_t *count, int64_t G, int32_t itemsize,
                       uint8_t *out, int64_t out_bytes);

/* The graph part of the arena image (GraphArena, _devicegraph.py; role of the
 * per-graph device copies of the reference, _octilegraph.py:170-189): blob g
 * of a packed batch goes to host + starts[g], and -- when ncls / ecls are
 * given -- the u8 label classes of its nodes (pad4) and nonzeros to the
 * cbytes[g] bytes in front of it.  ncls / ecls: the classes of all graphs
 * back to back. */
int gdh_assemble_arena(int64_t G, const uint8_t *blob, const int64_t *blob_off,
                       const int64_t *starts, const int64_t *cbytes, const int64_t *n_node,
                       const int64_t *n_nz, const uint8_t *ncls, const uint8_t *ecls,
                       uint8_t *host, int64_t host_bytes);

/* The job list of a kernel-matrix evaluation, (u32 i, u32 j) per pair in
 * row-major order: ny < 0: the upper triangle of an nx x nx symmetric matrix
 * with its diagonal, nx (nx + 1) / 2 jobs (i <= j); ny >= 0: all nx * ny
 * pairs (i, nx + j) of X against Y, whose graphs follow those of X in the
 * graph list.  Replaces the reference's Python loops over pairs
 * (graphdot/kernel/marginalized/_kernel.py:172-182). */
int gdh_pairwise_jobs(int64_t nx, int64_t ny, uint32_t *jobs);

#ifdef __cplusplus
}
#endif
#endif
