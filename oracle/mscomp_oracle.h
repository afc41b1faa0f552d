/* oracle/mscomp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, written from SURVEY.md section 8a, not copied) of the reference's
 * one-shot MS-XCA encoders. It is the checker for the HIP path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it. The product library (ms_compress_amd/csrc) never
 * links, loads or calls anything in this directory.
 *
 * Parity status: PINNED. There are no golden vectors in the reference's own tests, so the
 * restatement is pinned against outputs of the reference itself compiled here (oracle/_ref, built by
 * oracle/Makefile from the .cpp files under /root/reference/src) -- tests/test_oracle_vs_ref.py -- and against the
 * SURVEY 8c known-answer table + committed SHA-256 fixtures (tests/golden/).
 */
#ifndef MSCOMP_ORACLE_H
#define MSCOMP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_NONE = 0, ORC_LZNT1 = 2, ORC_XPRESS = 3, ORC_XPRESS_HUFF = 4 };
enum { ORC_OK = 0, ORC_ARG_ERROR = -2, ORC_DATA_ERROR = -3, ORC_MEM_ERROR = -4, ORC_BUF_ERROR = -5 };

/* ms_compress / ms_decompress / ms_max_compressed_size equivalents (mscomp.h:59,88,99). */
int    orc_compress(int format, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
int    orc_decompress(int format, const uint8_t* in, size_t in_len, uint8_t* out, size_t* out_len);
size_t orc_max_compressed_size(int format, size_t in_len);
int    orc_last_undefined(void);   /* 1 when the last orc_decompress of this thread met a stream on which the reference is undefined */

/* Component oracles (intermediate goldens so kernel stages can be diffed one by one). */
void orc_huff_lengths(const uint32_t counts[512], uint8_t lens[512]);            /* CreateCodes      */
void orc_huff_lengths_slow(const uint32_t counts[512], uint8_t lens[512]);       /* CreateCodesSlow  */
/* LZNT1Dictionary::Find at every position of one chunk (len 0 = no match). */
void orc_lznt1_match_table(const uint8_t* chunk, unsigned n, uint16_t* len, uint16_t* off);
/* XpressDictionary::Find at every position p < n-2 of a whole buffer (len 2 = no match); entries for
 * p >= n-2 are set to len 2. No lagging-fill rule here (that is parse state, not dictionary state). */
void orc_xpress_match_table(const uint8_t* buf, size_t n, uint32_t max_offset, uint32_t* len, uint32_t* off);

/* Multi-threaded "independent units" driver used for the cpu_baseline timing and for batch parity:
 * unit i = in[in_off[i] .. in_off[i+1]) compressed with one one-shot call into out[out_off[i] ..
 * out_off[i+1]) ; out_len[i] receives the size, status[i] the status. threads <= 0 -> 1. */
int orc_compress_units(int format, const uint8_t* in, const uint64_t* in_off, size_t n_units,
                       uint8_t* out, const uint64_t* out_off, uint64_t* out_len, int32_t* status, int threads);

/* `passes` passes of fn(format, unit i ...) over n_units independent units on `threads` threads; fn = any one-shot function (NULL:
 * orc_compress); returns the seconds spent. in_off / out_off have n_units + 1 entries (unit i = [off[i], off[i+1])). */
double orc_time_units(void* fn, int format, const uint8_t* in, const uint64_t* in_off, size_t n_units,
                      uint8_t* out, const uint64_t* out_off, uint64_t* out_len, int32_t* status, int threads, int passes);

/* The same with explicit unit lengths / capacities (units may alias each other in `in`) and ONE unit per grab, in the order given. */
double orc_time_units_ex(void* fn, int format, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, size_t n_units,
                         uint8_t* out, const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, int threads, int passes);
/* the same; *busy_seconds (may be NULL) = thread-seconds spent inside fn, summed over threads and passes of THIS call */
double orc_time_units_ex2(void* fn, int format, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, size_t n_units,
                          uint8_t* out, const uint64_t* out_off, const uint64_t* out_cap, uint64_t* out_len, int32_t* status, int threads, int passes,
                          double* busy_seconds);

/* LZNT1 dictionary flavour of orc_compress (process-wide, like the reference's build switch): 0 = the default dictionary
 * (include/mscomp/LZNT1Dictionary.h), 1 = the suffix-array dictionary (MSCOMP_WITH_LZNT1_SA_DICT, include/mscomp/LZNT1Dictionary_SA.h). */
void orc_set_lznt1_sa_dict(int on);

#ifdef __cplusplus
}
#endif
#endif
