/*
 * lva_decoder.h -- C ABI of the MI355X list-Viterbi decoder (liblva_hip.so).
 *
 * Drop-in boundary for ONE path of shubhamchandak94/nanopore_dna_storage: the
 * convolutional-code parallel list-Viterbi decode of a flappie transition-posterior
 * matrix.  The reference has no library API for this path -- its boundary is the
 * subprocess `viterbi_nanopore.out -m decode -i X.post -o OUT ...` -- so each entry
 * point below cites the reference code it replaces
 * (file viterbi/viterbi_convolutional_code.cpp unless stated otherwise).
 *
 * Plain C types only: pointers, sizes, POD structs.  No torch / HIP types.
 * All functions return LVA_OK (0) or a negative LVA_ERR_* code; none aborts.
 */
#ifndef LVA_DECODER_H
#define LVA_DECODER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LVA_OK 0
#define LVA_ERR_MEM_CONV (-1)        /* "Invalid mem_conv (allowed: 6, 8, 11, 14)"      :290-292 */
#define LVA_ERR_RATE (-2)            /* "Invalid rate parameter"                        :336-338 */
#define LVA_ERR_MSG_LEN (-3)         /* "Output length not even. Try padding ..."       :353-357 */
#define LVA_ERR_SYNC (-4)            /* sync marker too long / period short / bad char  :390-413 */
#define LVA_ERR_TOO_MANY_STATES (-5) /* runtime_error "Too many states"                 :595-597 */
#define LVA_ERR_POST_TOO_SHORT (-6)  /* runtime_error "Too small post matrix"           :600-601 */
#define LVA_ERR_MSG_TOO_LONG (-7)    /* runtime_error msg_len > BITSET_SIZE :604-605; uint8_t loop :831 */
#define LVA_ERR_NOMEM (-8)
#define LVA_ERR_HIP (-9)             /* a HIP runtime call failed (lva_last_hip_error has the text) */
#define LVA_ERR_ARG (-10)
#define LVA_ERR_NO_DEVICE (-11)      /* no gfx950 device visible: the product never falls back to CPU */
#define LVA_ERR_UNSUPPORTED (-12)

#define LVA_MAX_DEVIATION_DEFAULT 0xFFFFFFFFu /* reference default msg_len+mem_conv+1 (:238-240) */

typedef struct lva_decoder lva_decoder;

/* Decoder configuration = the decode-side CLI flags of the reference (:138-172):
 *   --mem-conv, -r/--rate, --msg-len, -l/--list-size, --max-deviation,
 *   --sync-marker, --sync-period.  (-t/--num-thr has no meaning on the GPU;
 *   --rc is per read, see lva_decode_batch.) */
typedef struct lva_config {
  int32_t mem_conv;
  int32_t rate;
  uint32_t msg_len;
  uint32_t list_size;
  uint32_t max_deviation;   /* LVA_MAX_DEVIATION_DEFAULT = unbanded */
  const char *sync_marker;  /* NULL or "" = none */
  uint32_t sync_period;
  int32_t device;           /* HIP device ordinal */
  int32_t max_slots;        /* reads in flight on the device; 0 = choose from free HBM */
  int32_t kernel;           /* 0 = default (4 where it is the faster one, else 2 where available, else 3, else 1);
                               1 = exact kernel, one thread per target; 2 = fast kernel + exact fix-up (L = 1, 2, 4, 8:
                               lva_step_fast / lva_step_acs; any other 2 <= L <= 64: lva_step_big); 3 = exact kernel, one
                               wavefront per target (2 <= L <= 64); 4 = fast kernel with lazy messages (L = 2, 4, 8:
                               messages materialised every second time step, lva_step_lazy).  Every mode gives the
                               reference's lists bit for bit on every input the reference decodes: targets the fast
                               kernels cannot decide (score ties, non-finite sums, fingerprint collisions) go through a
                               work list to an exact pass, and when that list overflows (tie-dense posteriors: quantised
                               or constant matrices) modes 2 and 4 redo the whole step on their exact path -- slower,
                               never refused (reference :762-796) */
  uint64_t mem_budget_bytes;/* cap on trellis memory; 0 = 60% of free HBM */
} lva_config;

/* Static facts about a code (set_conv_params :264-415). */
typedef struct lva_code_info {
  uint32_t nstate_pos, nstate_conv, oligo_len, msg_words;
  uint32_t initial_state, final_state;
  uint32_t g0, g1;
  int32_t pattern_len;
  uint8_t pattern[16];
} lva_code_info;

/* Counters of the last lva_decode_batch* call (bench.py's roofline object). */
typedef struct lva_profile {
  double step_kernel_ms;      /* HIP-event time from first to last trellis-step launch, on the decoder's stream */
  double total_ms;            /* HIP-event time of the whole call's device work (upload..results) */
  uint64_t step_launches;     /* trellis-step kernel launches */
  uint64_t read_steps;        /* sum over reads of nblk (one launch advances every active read one step) */
  double algorithmic_bytes;   /* sum over reads of sum_t [2 R(t) L (4+4W) + 160] */
  uint64_t fixup_states;      /* states redone by the exact kernel (kernel mode 2); a launch whose work list overflowed
                                 counts in overflow_steps instead (every state of the step was redone) */
  uint64_t fixup_reason[4];   /* of which: score tie on top, non-finite arithmetic, too many fingerprint
                                 matches, fingerprint collision */
  int32_t slots;              /* reads in flight */
  int32_t kernel;             /* kernel mode used */
  /* with lva_decoder_set_launch_events(d, 1): HIP events around every trellis-step launch */
  double dominant_kernel_ms;  /* sum over launches of the dominant kernel alone (lva_step_fast / lva_step_big / ...) */
  double step_pair_ms;        /* sum over launches of dominant kernel + fix-up pass */
  uint64_t timed_launches;    /* launches in those sums */
  double h2d_ms;              /* lva_decode_batch: HIP-event time of the host->device copy of the posteriors */
  uint64_t h2d_bytes;
  uint64_t overflow_steps;    /* launches whose work list of undecided targets overflowed: the exact path redid the whole
                                 step (tie-dense posteriors; correct, but a large slow-down the caller can now see) */
  double working_bytes;       /* algorithmic_bytes' sum over the band the kernels actually work on (lva_band_table's
                                 working_lo_hi): the bytes that are moved, about 5 % fewer at the benchmark shape */
} lva_profile;

/* "lva_hip <abi>.<n> (gfx950) build <id>": <id> is a hash of the library's source files taken when it was built
 * (csrc/Makefile), so that a profile or a counter file can name the library it was measured on (bench.py writes it into
 * every JSON line; profiles/ *_traffic.json carry it). */
const char *lva_version(void);
/* LVA_ABI_VERSION of the library that was loaded.  It changes whenever a struct of this header changes size or layout
 * (5: lva_profile gained overflow_steps and working_bytes): a caller built against another value must not pass structs. */
#define LVA_ABI_VERSION 5
int lva_abi_version(void);
const char *lva_strerror(int code);
const char *lva_last_hip_error(void);

/* --- host-only: code parameters and the encoder ------------------------------------------- */

/* set_conv_params (:264-415): validate and describe a code. */
int lva_code_describe(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc,
                      const char *sync_marker, uint32_t sync_period, lva_code_info *out);

/* The tables the kernels use, for inspection/tests.  Each output may be NULL.
 *   pos2msg[nstate_pos], ptype[nstate_pos], vmask[nstate_pos], vval[nstate_pos],
 *   predtab[4][nstate_conv] (uint16, zero rows for unused block types). */
int lva_code_tables(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc,
                    const char *sync_marker, uint32_t sync_period, uint32_t *pos2msg, uint8_t *ptype,
                    uint32_t *vmask, uint32_t *vval, uint16_t *predtab);

/* The band [lo, hi) of every time step of a read of nblk blocks, for inspection/tests (each output may be NULL, 2*nblk words):
 *   reference_lo_hi  as the reference computes it (:677-679, with the fused multiply-subtract of its build);
 *   working_lo_hi    what the kernels work on: the same without positions a path cannot have reached yet (> t + 1) and
 *                    positions that cannot reach the final one any more (< nstate_pos - nblk + t) -- their lists never
 *                    reach the output, so they are neither written nor read. */
int lva_band_table(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc, const char *sync_marker,
                   uint32_t sync_period, uint32_t nblk, uint32_t max_deviation, uint32_t *reference_lo_hi,
                   uint32_t *working_lo_hi);

/* `-m encode` (:215-225): conv_encode (:450-499) + 2-bit base packing (:540-551).
 * msgs: n_msgs*msg_len bytes of 0/1.  out_bases: n_msgs*oligo_len bytes, values 0..3 = A,C,G,T. */
int lva_encode(int32_t mem_conv, int32_t rate, uint32_t msg_len, const uint8_t *msgs, int32_t n_msgs,
               uint8_t *out_bases);

/* SURVEY 8(d) algorithmic bytes of one read of nblk blocks. */
int lva_algorithmic_bytes(int32_t mem_conv, int32_t rate, uint32_t msg_len, int32_t rc,
                          const char *sync_marker, uint32_t sync_period, uint32_t nblk,
                          uint32_t list_size, uint32_t max_deviation, double *out);

/* --- decoder ------------------------------------------------------------------------------ */

/* Replaces process start-up of `viterbi_nanopore.out -m decode` (main :137-214, set_conv_params,
 * table construction :624-650).  Fails with LVA_ERR_NO_DEVICE when no GPU is usable. */
int lva_decoder_create(const lva_config *cfg, lva_decoder **out);
void lva_decoder_destroy(lva_decoder *d);

/* Replaces read_crf_post (:553-575) + decode_post_conv_parallel_LVA (:589-858) + the list
 * writer (:248-253) for n_reads independent reads.
 *   post         host float32, all reads' .post matrices back to back, 40 floats per block
 *                (file order: rows 0-3 = into flip base b from state s at [b*8+s], row 4 = into flop);
 *   row_offsets  n_reads+1 block offsets into post (read i = blocks [off[i], off[i+1]));
 *   rc_flags     n_reads bytes, 1 = decode as reverse complement (--rc), NULL = all forward;
 *   out_msgs     n_reads*list_size*msg_len bytes of 0/1, best first (rows >= count untouched);
 *   out_scores   n_reads*list_size path scores (may be NULL);
 *   out_counts   n_reads: number of list entries (<= list_size), or a negative LVA_ERR_* for a
 *                read the reference would have thrown on (e.g. LVA_ERR_POST_TOO_SHORT). */
int lva_decode_batch(lva_decoder *d, const float *post, const int64_t *row_offsets, int32_t n_reads,
                     const uint8_t *rc_flags, uint8_t *out_msgs, float *out_scores, int32_t *out_counts);

/* Same, with `post` already resident in device memory (HBM) of the decoder's device. */
int lva_decode_batch_device(lva_decoder *d, const float *post_dev, const int64_t *row_offsets,
                            int32_t n_reads, const uint8_t *rc_flags, uint8_t *out_msgs,
                            float *out_scores, int32_t *out_counts);

/* As lva_decode_batch_device, but read i is the window of n_blocks[i] blocks that starts at block
 * first_block[i] of the resident buffer: decodes the payload windows found by
 * lva_locate_payload_batch_device in place.
 * replaces: helper.truncate_post_file (helper.py:212-224) + the decode call of generate_decoded_lists.py:80-89. */
int lva_decode_windows_device(lva_decoder *d, const float *post_dev, const int64_t *first_block, const int64_t *n_blocks,
                              int32_t n_reads, const uint8_t *rc_flags, uint8_t *out_msgs, float *out_scores,
                              int32_t *out_counts);

int lva_decoder_profile(const lva_decoder *d, lva_profile *out);
/* on != 0: record HIP events around every trellis-step launch of later decode calls (three per launch, on the
 * decoder's stream) so that lva_profile carries per-kernel times measured live; off by default. */
int lva_decoder_set_launch_events(lva_decoder *d, int32_t on);

/* Device helpers so that callers without a HIP binding (ctypes) can keep inputs resident. */
int lva_device_alloc(lva_decoder *d, uint64_t bytes, void **out_dev_ptr);
int lva_device_free(lva_decoder *d, void *dev_ptr);
int lva_device_upload(lva_decoder *d, void *dev_dst, const void *host_src, uint64_t bytes);
int lva_device_synchronize(lva_decoder *d);

/* ---------------------------------------------------------------------------------------------
 * the step in front of the list decoder on real data -- flappie's
 * flip-flop basecall of the posterior matrix and the barcode localisation on it, so that the
 * whole .post -> payload window -> decoded list chain stays on the device.
 * Posterior matrices are passed as for lva_decode_batch: one float32[blocks][40] buffer and
 * row_offsets[n_reads+1] in blocks (row_offsets[0] = 0).  A read may have at most 2^20 blocks (or called
 * bases) and a batch fewer than 2^31 blocks in all; beyond that the calls return LVA_ERR_ARG.
 * ------------------------------------------------------------------------------------------- */

/* Result of a barcode search for one read.
 * replaces: the tuple returned by helper.find_barcode_pos_in_post (helper.py:157-210) and the
 * orientation / length decision of generate_decoded_lists.py:68-79. */
typedef struct lva_payload_pos {
  int32_t start_pos, end_pos;   /* payload = blocks [start_pos, end_pos] of the read's matrix; -1, -1 on failure */
  int32_t dist_start, dist_end; /* edit distance of the best start / end barcode match; INT32_MAX = the reference's np.inf */
  int32_t rc;                   /* 1: the reverse-complement barcodes matched better (generate_decoded_lists.py:71-74) */
  int32_t ok;                   /* 0: "Failure in barcode removing." (generate_decoded_lists.py:76) */
} lva_payload_pos;

/* Basecall: the base string flappie writes as the second fastq line and the positions it writes
 * to --trans-output-file.
 * replaces: decode_crf_flipflop (flappie/src/decode.c:119-204), change_positions (decode.c:66-79)
 * and the loop of flappie/src/flappie.c:274-285.
 * Outputs of read i start at row_offsets[i] in bases_out / trans_out (capacity row_offsets[n_reads]
 * each; either may be NULL); nbases_out[i] = number of bases called. */
int lva_basecall_batch(lva_decoder *d, const float *post, const int64_t *row_offsets, int32_t n_reads,
                       char *bases_out, uint32_t *trans_out, int32_t *nbases_out);
int lva_basecall_batch_device(lva_decoder *d, const float *post_dev, const int64_t *row_offsets, int32_t n_reads,
                              char *bases_out, uint32_t *trans_out, int32_t *nbases_out);

/* Barcode search on given basecalls (one orientation).
 * replaces: helper.find_barcode_pos_in_post (helper.py:157-210); rc = 0, ok = (start_pos != -1).
 * bases / trans hold the basecall and the trans-file integers of read i at
 * [base_offsets[i], base_offsets[i+1]).  Barcodes: 1..64 characters. */
int lva_find_barcode_batch(lva_decoder *d, const char *bases, const uint32_t *trans, const int64_t *base_offsets,
                           int32_t n_reads, const char *start_barcode, const char *end_barcode, lva_payload_pos *out);

/* Basecall + barcode search in both orientations + the choice between them.
 * replaces: generate_decoded_lists.py:68-79 (START_BARCODE_RC = rc(end), END_BARCODE_RC = rc(start), :33-34);
 * min_len = MEM_CONV + MSG_LEN + 1 (:76).  The caller then decodes blocks [start_pos, end_pos]
 * with the rc flag (lva_decode_windows_device on the same resident buffer: no copy). */
int lva_locate_payload_batch(lva_decoder *d, const float *post, const int64_t *row_offsets, int32_t n_reads,
                             const char *start_barcode, const char *end_barcode, uint32_t min_len, lva_payload_pos *out);
int lva_locate_payload_batch_device(lva_decoder *d, const float *post_dev, const int64_t *row_offsets, int32_t n_reads,
                                    const char *start_barcode, const char *end_barcode, uint32_t min_len,
                                    lva_payload_pos *out);

/* ---------------------------------------------------------------------------------------------
 * the Reed-Solomon outer code, RS(65535, 65535 - redundancy) over GF(2^16)
 * (primitive polynomial x^16+x^12+x^3+x+1, generator roots alpha^0 .. alpha^(redundancy-1)) shortened to n_total
 * symbols -- one codeword per 16-bit column of the oligo payloads, all columns in one call.
 * Symbols are uint16 in host byte order = the little-endian 2-byte pieces the reference reads from its files
 * (RSCode_schifra/schifra_RS_16bit_fileio.cpp:89-103).  1 <= redundancy <= 4096, redundancy < n_total <= 65535.
 * pad_symbol = the value of the 65535 - n_total untransmitted leading symbols (the reference pads with ASCII '0'
 * bytes: 0x3030, RSCode_16bit_fileio.py:58,:104).
 * ------------------------------------------------------------------------------------------- */

/* replaces: RS_decode_16bit for every column (RSCode_schifra/RSCode_16bit_fileio.py:87-137) =
 * schifra decoder::decode(block, erasure_list) (RSCode_schifra/schifra_reed_solomon_decoder.hpp:64-168).
 *   symbols   [n_codewords][n_total] received words (the reference puts pad_symbol at erased positions, :242-250);
 *   erasures  n_erasures distinct positions in [0, n_total), shared by all codewords;
 *   out       [n_codewords][n_total - redundancy] decoded data symbols; where the reference's decoder gives up
 *             (no output file, :122-123) every symbol is fail_symbol (the reference: ASCII '0' bytes, 0x3030);
 *   ok        [n_codewords] 1 = decoded, 0 = given up (may be NULL). */
int lva_rs_decode(int32_t device, const uint16_t *symbols, int32_t n_codewords, int32_t n_total, int32_t redundancy,
                  const int32_t *erasures, int32_t n_erasures, uint16_t pad_symbol, uint16_t fail_symbol, uint16_t *out,
                  int32_t *ok);

/* replaces: RS_encode_16bit for every column (RSCode_16bit_fileio.py:48-77) = schifra encoder::encode
 * (schifra_reed_solomon_encoder.hpp:57-88): systematic; out [n_codewords][n_data + redundancy] = data, then parity. */
int lva_rs_encode(int32_t device, const uint16_t *data, int32_t n_codewords, int32_t n_data, int32_t redundancy,
                  uint16_t pad_symbol, uint16_t *out);
const char *lva_rs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
