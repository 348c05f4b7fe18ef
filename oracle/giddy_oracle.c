/*
 * giddy_oracle.c — CPU restatement of the reference decompression kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (libgiddy_amd/, include/)
 * links, loads or calls this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / the CPU timing leg.
 *
 * Parity status: the reference cannot be built here (needs nvcc, CUB, Boost and the
 * empty external/cuda-api-wrappers submodule — SURVEY.md §8c), so this oracle is
 * pinned against (a) the single known-answer test the reference ships, KAT-FOR-1
 * (examples/src/modes_of_use/common.cuh:31-49,91-102) and (b) independent numpy
 * statements committed as tests/golden/ fixtures.  For every scheme other than
 * Frame-of-Reference: PARITY UNPINNED by reference-held vectors.
 *
 * Every function restates element-level semantics (not thread choreography) of the
 * reference file:line cited above it.  Sizes are passed at run time; the bodies are
 * always_inline and dispatched with constant sizes so gcc -O3 specialises them
 * (that is what makes the cpu_baseline timing meaningful).
 *
 * "threads" argument: 1 = plain scalar loop; >1 = OpenMP parallel-for over
 * independent elements / anchored segments with that many threads.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define AINLINE static inline __attribute__((always_inline))
#define EXPORT __attribute__((visibility("default")))

/* ---- little-endian sized unsigned load/store (uint_t<N>, src/util/integer.h:194-210) ---- */
AINLINE uint64_t ld_u(const void* p, const int size, uint64_t i)
{
	switch (size) {
	case 1: return ((const uint8_t*) p)[i];
	case 2: return ((const uint16_t*)p)[i];
	case 4: return ((const uint32_t*)p)[i];
	default:return ((const uint64_t*)p)[i];
	}
}
AINLINE int64_t ld_s(const void* p, const int size, uint64_t i)
{
	switch (size) {
	case 1: return ((const int8_t*) p)[i];
	case 2: return ((const int16_t*)p)[i];
	case 4: return ((const int32_t*)p)[i];
	default:return ((const int64_t*)p)[i];
	}
}
AINLINE void st_u(void* p, const int size, uint64_t i, uint64_t v)
{
	switch (size) {
	case 1: ((uint8_t*) p)[i] = (uint8_t) v; break;
	case 2: ((uint16_t*)p)[i] = (uint16_t)v; break;
	case 4: ((uint32_t*)p)[i] = (uint32_t)v; break;
	default:((uint64_t*)p)[i] = v; break;
	}
}
AINLINE int64_t sext(uint64_t v, const int size)
{
	switch (size) {
	case 1: return (int8_t) v;
	case 2: return (int16_t)v;
	case 4: return (int32_t)v;
	default:return (int64_t)v;
	}
}
AINLINE uint64_t trunc_u(uint64_t v, const int size)
{
	return size >= 8 ? v : (v & ((((uint64_t)1) << (8 * size)) - 1));
}

#define OMP_FOR(threads) _Pragma("omp parallel for schedule(static) num_threads(threads) if((threads) > 1)")

/* ------------------------------------------------------------------------------------------
 * Parametric models — src/cuda/model_functors.hpp:76-111.
 *   constant: c[0]                            (:108-111)
 *   linear  : c[0] + c[1]*x, x converted to uint_t<ModelIndexSize>; usual arithmetic
 *             conversions make the sum unsigned; the return converts to T (:87-90).
 * Low bits of a two's-complement product/sum do not depend on the width it is formed in,
 * so the value is trunc_T(c0 + c1 * trunc_idx(x)) computed mod 2^64.
 * model_kind: 0 = constant, 1 = linear.  coefficients are signed, coef_size bytes each.
 * ------------------------------------------------------------------------------------------ */
AINLINE int64_t eval_model(const void* coefs, const int coef_size, const int model_kind,
                           uint64_t record, uint64_t x, const int model_index_size)
{
	const int dim = model_kind == 0 ? 1 : 2;
	uint64_t c0 = (uint64_t) ld_s(coefs, coef_size, record * dim);
	if (model_kind == 0) { return sext(trunc_u(c0, coef_size), coef_size); }
	uint64_t c1 = (uint64_t) ld_s(coefs, coef_size, record * dim + 1);
	uint64_t xi = trunc_u(x, model_index_size);
	return sext(trunc_u(c0 + c1 * xi, coef_size), coef_size);
}

/* ------------------------------------------------------------------------------------------
 * Frame of reference — src/kernels/decompression/frame_of_reference.cuh:49-79.
 *   out[p] = U( U(model_s(p - s*L)) + in[p] ),  s = p / L, in[] signed Compressed.
 * The oracle applies the `position >= length` guard (:55) to every position (the
 * reference's one-segment-per-block branch lacks it — SURVEY.md quirks register).
 * ------------------------------------------------------------------------------------------ */
AINLINE void for_body(void* out, const int usize, const void* in, const int csize,
                      const void* coefs, const int coef_size, const int model_kind,
                      uint64_t length, uint64_t seglen, const int model_index_size, int threads)
{
	uint64_t nseg = (length + seglen - 1) / seglen;
	OMP_FOR(threads)
	for (uint64_t s = 0; s < nseg; s++) {
		uint64_t start = s * seglen;
		uint64_t end = start + seglen < length ? start + seglen : length;
		for (uint64_t p = start; p < end; p++) {
			int64_t predicted = sext(trunc_u((uint64_t) eval_model(coefs, coef_size, model_kind, s, p - start, model_index_size), usize), usize);
			st_u(out, usize, p, (uint64_t)(predicted + ld_s(in, csize, p)));
		}
	}
}

EXPORT int oracle_frame_of_reference(void* out, int usize, const void* in, int csize,
	const void* coefs, int coef_size, int model_kind, uint64_t length, uint64_t seglen,
	int model_index_size, int threads)
{
	if (seglen == 0) return -1;
#define FOR_CASE(U, C) if (usize == U && csize == C) { for_body(out, U, in, C, coefs, coef_size, model_kind, length, seglen, model_index_size, threads); return 0; }
	FOR_CASE(2, 1) FOR_CASE(4, 1) FOR_CASE(4, 2) FOR_CASE(8, 1) FOR_CASE(8, 2) FOR_CASE(8, 4)
	for_body(out, usize, in, csize, coefs, coef_size, model_kind, length, seglen, model_index_size, threads);
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Model — src/kernels/decompression/model.cuh:44-52:  out[p] = U(model(p)), one record.
 * ------------------------------------------------------------------------------------------ */
EXPORT int oracle_model(void* out, int usize, const void* coefs, int coef_size, int model_kind,
	uint64_t length, int model_index_size, int threads)
{
	OMP_FOR(threads)
	for (uint64_t p = 0; p < length; p++) {
		st_u(out, usize, p, (uint64_t) eval_model(coefs, coef_size, model_kind, 0, p, model_index_size));
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Dictionary — src/kernels/decompression/dictionary.cuh:61-75 -> gather,
 * src/kernels/data_layout/gather.cuh:92-144:  out[p] = dict[key[p]] (raw bit copy).
 * ------------------------------------------------------------------------------------------ */
AINLINE void dict_body(void* out, const int usize, const void* keys, const int dsize,
                       const void* dict, uint64_t length, int threads)
{
	OMP_FOR(threads)
	for (uint64_t p = 0; p < length; p++) {
		st_u(out, usize, p, ld_u(dict, usize, ld_u(keys, dsize, p)));
	}
}
EXPORT int oracle_dictionary(void* out, int usize, const void* keys, int dsize, const void* dict,
	uint64_t length, int threads)
{
#define DICT_CASE(U, D) if (usize == U && dsize == D) { dict_body(out, U, keys, D, dict, length, threads); return 0; }
	DICT_CASE(1,1) DICT_CASE(2,1) DICT_CASE(2,2) DICT_CASE(4,1) DICT_CASE(4,2) DICT_CASE(4,4)
	DICT_CASE(8,1) DICT_CASE(8,2) DICT_CASE(8,4) DICT_CASE(8,8)
	return -1;
}

/* ------------------------------------------------------------------------------------------
 * Delta — src/kernels/decompression/delta.cuh:79-107 + src/kernels/reduction/scan.cuh:309-380:
 *   out[s*L + i] = anchor[s] + sum_{j<=i} zext(in[s*L + j])   (wrap at U bytes; inclusive
 *   scan seeded with the anchor; deltas are unsigned uint_t<CompressedSize>).
 * Stated for every L (the reference is only right for L >= blockDim — quirk a3).
 * ------------------------------------------------------------------------------------------ */
AINLINE void delta_body(void* out, const int usize, const void* in, const int csize,
                        const void* anchors, uint64_t length, uint64_t seglen, int threads)
{
	uint64_t nseg = (length + seglen - 1) / seglen;
	OMP_FOR(threads)
	for (uint64_t s = 0; s < nseg; s++) {
		uint64_t start = s * seglen;
		uint64_t end = start + seglen < length ? start + seglen : length;
		uint64_t acc = ld_u(anchors, usize, s);
		for (uint64_t p = start; p < end; p++) {
			acc += ld_u(in, csize, p);
			st_u(out, usize, p, acc);
		}
	}
}
EXPORT int oracle_delta(void* out, int usize, const void* in, int csize, const void* anchors,
	uint64_t length, uint64_t seglen, int threads)
{
	if (seglen == 0) return -1;
#define DELTA_CASE(U, C) if (usize == U && csize == C) { delta_body(out, U, in, C, anchors, length, seglen, threads); return 0; }
	DELTA_CASE(2,1) DELTA_CASE(4,1) DELTA_CASE(4,2) DELTA_CASE(8,1) DELTA_CASE(8,2) DELTA_CASE(8,4)
	return -1;
}

/* ------------------------------------------------------------------------------------------
 * Discard-zero-bytes, fixed width — src/kernels/decompression/discard_zero_bytes_fixed.cuh:54-76,
 * :108-124:  out[p] = zext(in[p]) (little-endian placement; U > C).
 * ------------------------------------------------------------------------------------------ */
AINLINE void dzbf_body(void* out, const int usize, const void* in, const int csize, uint64_t length, int threads)
{
	OMP_FOR(threads)
	for (uint64_t p = 0; p < length; p++) { st_u(out, usize, p, ld_u(in, csize, p)); }
}
EXPORT int oracle_dzb_fixed(void* out, int usize, const void* in, int csize, uint64_t length, int threads)
{
	if (!(usize > csize)) return -1;
#define DZBF_CASE(U, C) if (usize == U && csize == C) { dzbf_body(out, U, in, C, length, threads); return 0; }
	DZBF_CASE(2,1) DZBF_CASE(4,1) DZBF_CASE(4,2) DZBF_CASE(8,1) DZBF_CASE(8,2) DZBF_CASE(8,4)
	return -1;
}

/* ------------------------------------------------------------------------------------------
 * Discard-zero-bytes, variable width — src/kernels/decompression/discard_zero_bytes_variable.cuh
 *   get_element_size :44-58 (+ bit_subsequence, src/cuda/bit_operations.cuh:344-362),
 *   pad_high_bytes :196-213, warp body :276-329, kernel :397-438.
 *   per  = floor(8*container_size / bits)
 *   f_i  = (sizes[i / per] >> ((i % per) * bits)) & (2^bits - 1);  n_i = min_size + f_i
 *   off_i = anchors[s] + sum_{j = s*L}^{i-1} n_j;   out[i] = little-endian n_i bytes at off_i
 * Stated for every tail length (the reference's last-warp test is quirk a8).
 * ------------------------------------------------------------------------------------------ */
EXPORT int oracle_dzb_variable(void* out, int usize, const uint8_t* data, const void* sizes,
	int container_size, const void* anchors, int index_size, uint64_t seglen, uint64_t length,
	unsigned min_size, unsigned bits, int threads)
{
	if (seglen == 0 || bits == 0 || bits > 8u * container_size) return -1;
	uint64_t per = (8u * container_size) / bits;
	uint64_t mask = bits >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << bits) - 1);
	uint64_t nseg = (length + seglen - 1) / seglen;
	int status = 0;
	OMP_FOR(threads)
	for (uint64_t s = 0; s < nseg; s++) {
		uint64_t start = s * seglen;
		uint64_t end = start + seglen < length ? start + seglen : length;
		uint64_t off = ld_u(anchors, index_size, s);
		for (uint64_t i = start; i < end; i++) {
			uint64_t container = ld_u(sizes, container_size, i / per);
			uint64_t n = min_size + ((container >> ((i % per) * bits)) & mask);
			if (n > (uint64_t) usize) { status = -2; n = usize; }
			uint64_t v = 0;
			for (uint64_t k = 0; k < n; k++) { v |= ((uint64_t) data[off + k]) << (8 * k); }
			st_u(out, usize, i, v);
			off += n;
		}
	}
	return status;
}

/* ------------------------------------------------------------------------------------------
 * Incidence bitmaps — src/kernels/decompression/incidence_bitmaps.cuh:47-60 (bit peel,
 * ascending bitmap index => highest index wins), :94-126 (layout, dictionary translation).
 *   A = round_up(bitmap_length, alignment) words between consecutive bitmaps
 *   out[32*w + j] = dict[b] (or U(b)) for the bitmap b with bit j of bitmaps[b*A + w] set.
 * Positions covered by no bitmap are undefined in the reference (stale shared memory);
 * the oracle leaves them untouched and reports how many there were.
 * ------------------------------------------------------------------------------------------ */
EXPORT int64_t oracle_incidence_bitmaps(void* out, int usize, const uint32_t* bitmaps,
	const void* dict, uint64_t bitmap_length, uint32_t num_bitmaps, uint32_t alignment,
	int use_dictionary, int threads)
{
	uint64_t A = ((bitmap_length + alignment - 1) / alignment) * alignment;
	int64_t uncovered = 0;
	_Pragma("omp parallel for schedule(static) num_threads(threads) if(threads > 1) reduction(+:uncovered)")
	for (uint64_t w = 0; w < bitmap_length; w++) {
		for (unsigned j = 0; j < 32; j++) {
			int found = -1;
			for (uint32_t b = 0; b < num_bitmaps; b++) {
				if ((bitmaps[(uint64_t) b * A + w] >> j) & 1u) { found = (int) b; }
			}
			if (found < 0) { uncovered++; continue; }
			st_u(out, usize, 32 * w + j, use_dictionary ? ld_u(dict, usize, (uint64_t) found) : (uint64_t) found);
		}
	}
	return uncovered;
}

/* ------------------------------------------------------------------------------------------
 * Run-length encoding — src/kernels/decompression/run_length_encoding.cuh
 *   per-segment parameters :284-325, run walk :129-157.
 *   segment s (outputs [s*L, min((s+1)*L, N))) starts inside run anchors[s] at offset
 *   intra[s]; out[p] = run_data[r(p)].
 * The oracle walks from each segment's anchor (so the anchor arrays are exercised),
 * not from the start of the column.
 * ------------------------------------------------------------------------------------------ */
AINLINE void rle_body(void* out, const int usize, const void* run_data, const void* run_lengths,
	const int lsize, const void* anchors, const int asize, const void* intra, uint64_t seglen,
	uint64_t num_segments, uint64_t num_runs, uint64_t length, int threads, int* status)
{
	OMP_FOR(threads)
	for (uint64_t s = 0; s < num_segments; s++) {
		uint64_t start = s * seglen;
		uint64_t end = start + seglen < length ? start + seglen : length;
		uint64_t r = ld_u(anchors, asize, s);
		uint64_t remaining = ld_u(run_lengths, lsize, r) - ld_u(intra, lsize, s);
		uint64_t v = ld_u(run_data, usize, r);
		for (uint64_t p = start; p < end; p++) {
			while (remaining == 0) {
				r++;
				if (r >= num_runs) { *status = -2; r = num_runs - 1; remaining = ~(uint64_t)0; break; }
				remaining = ld_u(run_lengths, lsize, r);
				v = ld_u(run_data, usize, r);
			}
			st_u(out, usize, p, v);
			remaining--;
		}
	}
}
EXPORT int oracle_run_length_encoding(void* out, int usize, const void* run_data,
	const void* run_lengths, int lsize, const void* anchors, int asize, const void* intra,
	uint64_t seglen, uint64_t num_segments, uint64_t num_runs, uint64_t length, int threads)
{
	int status = 0;
	if (seglen == 0) return -1;
#define RLE_CASE(U, L_) if (usize == U && lsize == L_) { rle_body(out, U, run_data, run_lengths, L_, anchors, asize, intra, seglen, num_segments, num_runs, length, threads, &status); return status; }
	RLE_CASE(1,1) RLE_CASE(2,1) RLE_CASE(4,1) RLE_CASE(8,1)
	RLE_CASE(1,2) RLE_CASE(2,2) RLE_CASE(4,2) RLE_CASE(8,2)
	RLE_CASE(1,4) RLE_CASE(2,4) RLE_CASE(4,4) RLE_CASE(8,4)
	RLE_CASE(1,8) RLE_CASE(4,8) RLE_CASE(8,8)
	return -1;
}

/* ------------------------------------------------------------------------------------------
 * Run-position encoding — src/kernels/decompression/run_position_encoding.cuh
 *   per-segment parameters :265-307, run search :111-150.
 *   start(r) absolute, or relative to the start of the segment the run starts in
 *   (relative != 0).  anchors[s] = index of the run covering position s*L.
 *   out[p] = run_data[r(p)], r(p) = last run with start(r) <= p.
 * Semantic statement: a run that starts exactly on a segment boundary belongs to the
 * next segment only (the reference mis-handles that case in relative mode — quirk a7).
 * ------------------------------------------------------------------------------------------ */
AINLINE void rpe_body(void* out, const int usize, const void* run_data, const void* positions,
	const int psize, const int relative, const void* anchors, const int asize, uint64_t seglen,
	uint64_t num_anchors, uint64_t num_runs, uint64_t length, int threads)
{
	OMP_FOR(threads)
	for (uint64_t s = 0; s < num_anchors; s++) {
		uint64_t start = s * seglen;
		uint64_t end = start + seglen < length ? start + seglen : length;
		uint64_t r = ld_u(anchors, asize, s);
		/* runs after r that start inside this segment: r+1 .. hi (inclusive) */
		uint64_t hi = (s + 1 == num_anchors) ? num_runs - 1 : ld_u(anchors, asize, s + 1);
		uint64_t v = ld_u(run_data, usize, r);
		uint64_t next = r + 1;
		uint64_t next_start = ~(uint64_t)0;
		#define RPE_FETCH_NEXT() do { \
			next_start = ~(uint64_t)0; \
			if (next <= hi) { \
				uint64_t pos_ = ld_u(positions, psize, next); \
				uint64_t abs_ = relative ? start + pos_ : pos_; \
				/* a relative position of 0 (or an absolute one at/after `end`) is a run of the next segment */ \
				if (!(relative && pos_ == 0) && abs_ < start + seglen) { next_start = abs_; } \
			} } while (0)
		RPE_FETCH_NEXT();
		for (uint64_t p = start; p < end; p++) {
			while (p >= next_start) {
				r = next; v = ld_u(run_data, usize, r); next++;
				RPE_FETCH_NEXT();
			}
			st_u(out, usize, p, v);
		}
	}
}
EXPORT int oracle_run_position_encoding(void* out, int usize, const void* run_data,
	const void* positions, int psize, int relative, const void* anchors, int asize,
	uint64_t seglen, uint64_t num_anchors, uint64_t num_runs, uint64_t length, int threads)
{
	if (seglen == 0) return -1;
#define RPE_CASE(U, P) if (usize == U && psize == P) { rpe_body(out, U, run_data, positions, P, relative, anchors, asize, seglen, num_anchors, num_runs, length, threads); return 0; }
	RPE_CASE(1,2) RPE_CASE(2,2) RPE_CASE(4,2) RPE_CASE(8,2)
	RPE_CASE(1,4) RPE_CASE(2,4) RPE_CASE(4,4) RPE_CASE(8,4) RPE_CASE(4,8)
	rpe_body(out, usize, run_data, positions, psize, relative, anchors, asize, seglen, num_anchors, num_runs, length, threads);
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Naive scatter (patching) — src/kernels/data_layout/scatter.cuh:42-45:
 *   target[idx[i]] = data[i];  distinct indices required.
 * ------------------------------------------------------------------------------------------ */
EXPORT int oracle_scatter(void* target, int esize, const void* data, const void* indices,
	int isize, uint64_t data_length, int threads)
{
	OMP_FOR(threads)
	for (uint64_t i = 0; i < data_length; i++) {
		st_u(target, esize, ld_u(indices, isize, i), ld_u(data, esize, i));
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Compressed-indices scatter — src/kernels/data_layout/compressed_indices_scatter.cuh
 *   single run :40-53, sequential run loop :83-124, per-anchor block :236-269.
 *   For anchor block a (patches [a*P, min((a+1)*P, n))): runs anchors[a] .. anchors[a+1]-1
 *   (last block: .. num_runs-1), cursor k starts at a*P; for t < run_lengths[rho]:
 *     target[baseline[rho] + off_rho[t]] = data[k + t],   off_rho = uintW[] at
 *     offset_bytes + start_pos[rho] * OutputIndexSize,  W = offset_sizes[rho] in BITS;
 *   W not in {8,16,32,64}: run skipped (its patches still advance the cursor) (:105-122).
 * ------------------------------------------------------------------------------------------ */
EXPORT int oracle_compressed_indices_scatter(void* target, int esize, const void* data,
	const void* run_lengths, int lsize, const uint8_t* offset_sizes, const void* baselines,
	int osize, const void* start_pos, int isize, const uint8_t* offset_bytes,
	const void* anchors, uint64_t anchoring_period, uint64_t num_runs, uint64_t data_length,
	int threads)
{
	if (data_length == 0 || anchoring_period == 0) return -1;
	uint64_t num_blocks = (data_length + anchoring_period - 1) / anchoring_period;
	OMP_FOR(threads)
	for (uint64_t a = 0; a < num_blocks; a++) {
		uint64_t first = ld_u(anchors, isize, a);
		int last_block = ((a + 1) * anchoring_period >= data_length);
		uint64_t nruns = (last_block ? num_runs : ld_u(anchors, isize, a + 1)) - first;
		uint64_t k = a * anchoring_period;
		for (uint64_t rho = first; rho < first + nruns; rho++) {
			uint64_t len = ld_u(run_lengths, lsize, rho);
			uint64_t base = ld_u(baselines, osize, rho);
			unsigned W = offset_sizes[rho];
			const uint8_t* offs = offset_bytes + ld_u(start_pos, isize, rho) * (uint64_t) osize;
			if (W == 8 || W == 16 || W == 32 || W == 64) {
				for (uint64_t t = 0; t < len; t++) {
					uint64_t o = 0;
					memcpy(&o, offs + t * (W / 8), W / 8);
					st_u(target, esize, trunc_u(base + o, osize), ld_u(data, esize, k + t));
				}
			}
			k += len;
		}
	}
	return 0;
}

EXPORT int oracle_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
