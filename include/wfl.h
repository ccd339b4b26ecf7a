/*
 * wfl.h -- C ABI of libwfl.so, the MI355X-native differentiable-WFST loss engine.
 *
 * This is the drop-in boundary for the hot path of facebookresearch/gtn_applications:
 * the criterion layer (criterions/{ctc,asg,stc,transducer}.py) reaches its arithmetic through the
 * pybind11 bindings of the external `gtn` C++ library.  Every entry point below names the gtn call
 * sites (file:line under /root/reference) whose work it replaces.  There is no reference header to
 * mirror (gtn is not vendored), so the ABI is designed for the path:
 *
 *   - plain C symbols, plain pointers and sizes, no C++/torch types, no exceptions;
 *   - every function returns an int status (WFL_OK == 0); wfl_last_error() gives a thread-local
 *     message for the last failure on the calling thread;
 *   - device entry points take caller-owned DEVICE buffers and a hipStream_t (passed as void*),
 *     enqueue work asynchronously and never synchronise, allocate or free device memory;
 *   - host graph entry points (wfl_graph_*) work on opaque handles and never touch the GPU;
 *   - no hidden global state: re-entrant from several host threads (autograd worker threads).  What a step remembers
 *     between calls (the CTC step's choice of launch) lives in memory the caller hands in (wfl_ctc_call); the gradient
 *     beside the lattice sweeps keeps per-device COUNTERS of whether kernels of two streams overlap on this stack
 *     (wfl_lattice_diagnostics) -- they pick between two launch plans with identical results.
 *
 * Emissions are always float32 [B, T, C] row-major ("the emissions graph": gtn.linear_graph +
 * set_weights, ctc.py:40-44, asg.py:96-100, stc.py:74-78, transducer.py:262-264 -- here the
 * tensor IS the graph, nothing is materialised).
 */
#ifndef WFL_H_
#define WFL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFL_OK 0
#define WFL_ERR_INVALID 1     /* bad argument (shape, null pointer, label out of range ...) */
#define WFL_ERR_UNSUPPORTED 2 /* valid request this build cannot serve (e.g. lattice too large for LDS) */
#define WFL_ERR_RUNTIME 3     /* HIP runtime error (message has the hipError string) */

#define WFL_EPSILON (-1) /* gtn.epsilon; pinned by tests/trans_backoff_test.txt:3 */

#define WFL_SEMIRING_LOG 0      /* forward_score  */
#define WFL_SEMIRING_TROPICAL 1 /* viterbi_score / viterbi_path */

const char* wfl_last_error(void);
int wfl_version(void);

/* ------------------------------------------------------------------------------------------------
 * Host WFST library (replaces gtn.Graph and the graph functions the criteria call)
 * ------------------------------------------------------------------------------------------------ */
typedef struct wfl_graph wfl_graph; /* opaque */

/* gtn.Graph(calc_grad) -- ctc.py:16, asg.py:57,73, stc.py:34, transducer.py:16,24,33,65,87,352 */
wfl_graph* wfl_graph_new(void);
void wfl_graph_free(wfl_graph* g);
wfl_graph* wfl_graph_clone(const wfl_graph* g);
/* Graph.add_node(start, accept) -> node id */
int wfl_graph_add_node(wfl_graph* g, int start, int accept);
/* Graph.add_arc(src, dst, ilabel, olabel, weight) -> arc id (insertion order), <0 on error */
int wfl_graph_add_arc(wfl_graph* g, int src, int dst, int ilabel, int olabel, float weight);
/* bulk forms of the two calls above (same ordering semantics) */
int wfl_graph_add_nodes(wfl_graph* g, int n, const uint8_t* start, const uint8_t* accept);
int wfl_graph_add_arcs(wfl_graph* g, int64_t n, const int32_t* src, const int32_t* dst,
                       const int32_t* ilabel, const int32_t* olabel, const float* weight);
int wfl_graph_num_nodes(const wfl_graph* g);
int64_t wfl_graph_num_arcs(const wfl_graph* g);
/* copy out: any pointer may be NULL.  start/accept: [num_nodes] bytes; arcs: [num_arcs] each */
int wfl_graph_get(const wfl_graph* g, uint8_t* start, uint8_t* accept, int32_t* src, int32_t* dst,
                  int32_t* ilabel, int32_t* olabel, float* weight);
/* Graph.set_weights(ptr): num_arcs float32 in arc-id order (ctc.py:44, asg.py:66, transducer.py:256) */
int wfl_graph_set_weights(wfl_graph* g, const float* w);
/* Graph.arc_sort(olabel): orders each node's in/out arc lists by label; arc ids are unchanged */
int wfl_graph_arc_sort(wfl_graph* g, int olabel);
/* gtn.compose / gtn.intersect (ctc.py:50, asg.py:112-114, stc.py:86, transducer.py:216-287):
 * first.olabel is matched with second.ilabel, epsilons advance alone, weights add, result trimmed.
 * If prov_first / prov_second are non-NULL they receive malloc'ed arrays [num_arcs(result)] with
 * the arc id each result arc came from (or -1); release them with wfl_free(). */
wfl_graph* wfl_graph_compose(const wfl_graph* first, const wfl_graph* second, int32_t** prov_first,
                             int32_t** prov_second);
/* gtn.remove(g, label) (transducer.py:221,229,269,274); prov as above (arc id in g) */
wfl_graph* wfl_graph_remove(const wfl_graph* g, int ilabel, int olabel, int32_t** prov);
/* gtn.project_input / project_output (transducer.py:229,269,273) */
wfl_graph* wfl_graph_project(const wfl_graph* g, int output);
/* gtn.viterbi_path on a host graph (transducer.py:228: tiny path (x) token-graph compositions).
 * Ties: maximum weight, then fewest non-epsilon output labels ("we take the shortest",
 * transducer.py:226-227), then first found.  Result is a chain graph. */
wfl_graph* wfl_graph_viterbi_path(const wfl_graph* g);
/* project_input(remove(compose(tokens, tokens_target))) (transducer.py:273-276) without the composition, for
 * tokens = make_token_graph(N, blank="optional", allow_repeats=False) (transducer.py:78-123) and a
 * single-start acceptor over the tokens: isomorphic to what the three calls give.  NULL (and no
 * error) if `tokens` does not have that shape -- use the three calls. */
wfl_graph* wfl_graph_token_alignments(const wfl_graph* tokens, const wfl_graph* tokens_target);
/* gtn.equal / gtn.isomorphic (tests/transducer_test.py:47-55,376-418) */
int wfl_graph_equal(const wfl_graph* a, const wfl_graph* b);
int wfl_graph_isomorphic(const wfl_graph* a, const wfl_graph* b);
/* gtn.loadtxt / savetxt text format (utils.py:261, build_transitions.py:221; format pinned by
 * tests/trans_backoff_test.txt) */
wfl_graph* wfl_graph_loadtxt(const char* path);
int wfl_graph_savetxt(const wfl_graph* g, const char* path);
/* gtn.load / gtn.save (utils.py:261 reads config["transitions"] with gtn.load; scripts/build_transitions.py:221
 * writes it with gtn.save).  load sniffs the file: gtn text (as above) or gtn's binary layout -- int32 counts
 * {nodes, start, accept, arcs}, start ids, accept ids, then {src, dst, ilabel, olabel, float weight} per arc --
 * restated from gtn's published utils.cpp and UNPINNED (gtn is not vendored): the reader cross-checks the counts
 * against the file size and id ranges (both plausible count orders) and fails loudly on anything else. */
wfl_graph* wfl_graph_load(const char* path);
int wfl_graph_save(const wfl_graph* g, const char* path);
void wfl_free(void* p);

/* ------------------------------------------------------------------------------------------------
 * Packed lattice batch: B acceptors A_b ready for the device kernels.
 *
 * Per utterance b (all indices local to b): Q_b states renumbered by epsilon level, A_b labelled
 * arcs sorted by destination, E_b epsilon arcs sorted by destination.  One int32 blob and one
 * float blob hold everything; `wfl_lattice_desc` says where each array starts (element offsets).
 * wfl_lattice_pack*() build the blobs on the host; the caller uploads them (one H2D each) and
 * passes device pointers + the descriptor (by value, host memory) to the kernels.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t B, max_states, max_arcs, max_eps, max_labels, max_levels;
  int64_t total_states, total_arcs, total_eps, total_labels;
  int32_t shared; /* 1: one graph shared by every utterance (transition graphs) */
  /* element offsets into the int32 blob */
  int64_t state_off; /* [B+1] first state of b in the per-state arrays                         */
  int64_t arc_off;   /* [B+1] first labelled arc of b                                         */
  int64_t eps_off;   /* [B+1] first epsilon arc of b                                          */
  int64_t lab_off;   /* [B+1] first entry of b in `labels`                                    */
  int64_t lvl_off;   /* [B+1] first entry of b in `lvl_ptr` (n_levels_b + 1 entries each)      */
  int64_t in_ptr;    /* [total_states + B] CSR by destination over labelled arcs (local ids)   */
  int64_t out_ptr;   /* [total_states + B] CSR by source                                      */
  int64_t out_arc;   /* [total_arcs] labelled-arc ids in by-source order                      */
  int64_t ein_ptr;   /* [total_states + B] CSR by destination over epsilon arcs               */
  int64_t eout_ptr;  /* [total_states + B]                                                    */
  int64_t eout_arc;  /* [total_eps]                                                           */
  int64_t arc_src, arc_dst, arc_slot, arc_lab, arc_wid; /* [total_arcs] each                   */
  int64_t eps_src, eps_dst, eps_wid;                    /* [total_eps] each                    */
  int64_t labels;    /* [total_labels] distinct emission columns used by b (slot -> column)    */
  int64_t lvl_ptr;   /* level boundaries (state ranges) for the epsilon closure                */
  int64_t arc_orig;  /* [total_arcs] caller's arc id of each labelled arc (for Viterbi paths)  */
  int64_t eps_orig;  /* [total_eps]                                                           */
  int64_t slot_ptr;  /* [total_labels + B] CSR by emission slot over labelled arcs (gradient rows) */
  int64_t slot_arc;  /* [total_arcs] labelled-arc ids in by-slot order                         */
  int64_t int_words; /* size of the int32 blob                                                 */
  /* element offsets into the float blob */
  int64_t arc_w;     /* [total_arcs] constant weight (NaN is stored as -inf)    */
  int64_t eps_w;     /* [total_eps]                                                           */
  int64_t start_w;   /* [total_states] 0 for start states else -inf                           */
  int64_t accept_w;  /* [total_states] 0 for accept states else -inf                          */
  int64_t float_words;
} wfl_lattice_desc;

typedef struct wfl_lattice_host wfl_lattice_host; /* opaque: descriptor + host blobs */

/* Generic packer: one graph per utterance (or a single shared graph, shared=1, n_graphs=1).
 * graphs[b] must be an acceptor over emission columns: arc ilabel in [0,C) or WFL_EPSILON.
 * wid[b] (may be NULL) gives, per arc of graphs[b], the index of the learnable weight it adds
 * (-1: none).  Replaces what gtn.intersect(emissions, A_b) needs to know about A_b. */
wfl_lattice_host* wfl_lattice_pack(const wfl_graph* const* graphs, const int32_t* const* wid,
                                   int n_graphs, int B, int shared, int C);
/* Hint that a batch is about to be packed: wakes the host pool's sleeping workers so that the job
 * submitted a few tens of microseconds later finds them polling (no-op when the pool does not
 * spin, WFL_HOST_SPIN_US=0). */
void wfl_host_pool_wake(void);

/* TransducerLossFunction.forward's per-sample host work for a whole batch (transducer.py:262-281 under
 * gtn.parallel_for, :296): for every target b (flat int32 graphemes + offsets[B+1])
 *     tokens_target = remove(project_output(compose(chain(target_b), lexicon)))
 *     alignments_b  = project_input(remove(compose(tokens, tokens_target)))
 *     [alignments_b = compose(transitions, alignments_b); wid = arc of `transitions` behind each arc]
 * on a persistent pool of host threads (nthreads: 1 = serial in the caller, otherwise the pool: one thread per
 * host core up to 64), packed like wfl_lattice_pack.  With `transitions` the arcs' constant weights are 0 (the
 * learnable weights are added on the device through wid), as the reference overwrites them (transducer.py:255). */
wfl_lattice_host* wfl_transducer_pack_batch(const wfl_graph* tokens, const wfl_graph* lexicon,
                                            const wfl_graph* transitions, const int32_t* targets,
                                            const int64_t* offsets, int B, int C, int nthreads);
/* The same, with the blobs written to the CALLER's buffer (typically pinned staging memory that is then uploaded:
 * saves the copy out of the handle) if they fit: `dst` (16-byte aligned, dst_bytes) receives
 * [float blob | reserve_floats floats left for the caller | pad to 16 bytes | int32 blob].
 * wfl_lattice_host_external(h) then returns the byte offset of the int blob in dst (and the handle holds only the
 * descriptor); -1 means the blobs did not fit (or dst was NULL) and are in the handle as usual. */
wfl_lattice_host* wfl_transducer_pack_batch_into(const wfl_graph* tokens, const wfl_graph* lexicon,
                                                 const wfl_graph* transitions, const int32_t* targets,
                                                 const int64_t* offsets, int B, int C, int nthreads, void* dst,
                                                 int64_t dst_bytes, int64_t reserve_floats);
int64_t wfl_lattice_host_external(const wfl_lattice_host* h);
/* Transducer.viterbi's decode stage for a whole batch -- transducer.py:221-232, the part of process(b) behind the best
 * frame path, run under gtn.parallel_for (transducer.py:232) and called in every training step (train.py:278-279):
 *     out_b = labels_to_list(remove(project_output(viterbi_path(compose(chain(labels_b), tokens)))))
 * labels: the frame-level label paths, flat, utterance b = labels[offsets[b] .. offsets[b+1]) (back-off epsilons already
 * removed).  out (caller's buffer, out_capacity int32; offsets[B] - offsets[0] always suffices for make_token_graph
 * graphs) receives the token sequences back to back, out_offsets[B+1] their boundaries; an utterance without an
 * accepting path decodes to nothing.  Ties (allow_repeats) go to the path with the fewest output labels, as
 * wfl_graph_viterbi_path breaks them.  The graphs make_token_graph builds (transducer.py:78-123) are recognised arc
 * for arc and decoded directly -- collapse repeated labels, drop blanks, within what the graph accepts --; any other
 * `tokens` goes through the graph algebra per utterance on the host pool (nthreads: 1 = serial in the caller). */
int wfl_transducer_decode_batch(const wfl_graph* tokens, const int32_t* labels, const int64_t* offsets, int B,
                                int32_t* out, int64_t out_capacity, int64_t* out_offsets, int nthreads);
/* Bulk builders for the three fixed-topology label graphs (no per-arc host calls):
 *   CTC  create_ctc_graph          ctc.py:15-29     targets flat + offsets[B+1], blank
 *   ASG  create_force_align_graph  asg.py:72-81 composed with the transitions graph asg.py:54-69:
 *        arcs carry wid into W[(C+1),C] row-major
 *   STC  create_stc_graph          stc.py:23-64     star_idx, log(prob) on star arcs */
wfl_lattice_host* wfl_lattice_pack_ctc(const int32_t* targets, const int64_t* offsets, int B, int blank, int C);
wfl_lattice_host* wfl_lattice_pack_asg_fal(const int32_t* targets, const int64_t* offsets, int B, int C);
wfl_lattice_host* wfl_lattice_pack_stc(const int32_t* targets, const int64_t* offsets, int B, int star_idx,
                                       float log_prob, int C);
void wfl_lattice_host_free(wfl_lattice_host* h);
const wfl_lattice_desc* wfl_lattice_host_desc(const wfl_lattice_host* h);
const int32_t* wfl_lattice_host_ints(const wfl_lattice_host* h);
const float* wfl_lattice_host_floats(const wfl_lattice_host* h);

/* ------------------------------------------------------------------------------------------------
 * Device kernels: generic lattice engine
 *   loss_b-related quantity logZ_b = forward_score(intersect(emissions_b, A_b))
 * ------------------------------------------------------------------------------------------------ */
/* Bytes of scratch the calls below need: xg [B,T,max_labels], alpha/beta [sum_b (T+1) Q_b].
 * Returned through the out pointers (element counts of float32). */
int wfl_lattice_workspace(const wfl_lattice_desc* d, int T, int64_t* xg_elems, int64_t* ab_elems);
/* Diagnostics: offset (float32 elements) in the alpha buffer of the int32 [B] array that says how each
 * utterance of a log-semiring wfl_lattice_forward was swept (1: fp64 probability domain, 0: fp32 log
 * domain -- acceptor shape, or the certificate's repair). */
int wfl_lattice_formats_offset(const wfl_lattice_desc* d, int T, int64_t* offset);

/* Stage 1 (all CUs, coalesced): xg[b,t,k] = x[b,t,labels_b[k]].
 * If row_lse != NULL it also receives logsumexp_c x[b,t,c] and xg is written log-softmaxed
 * (fused torch.nn.functional.log_softmax of ctc.py:107 / transducer.py:186-187). */
int wfl_lattice_gather(const wfl_lattice_desc* d, const int32_t* ints, const float* x, int T, int C,
                       float* xg, float* row_lse, void* stream);

/* Stage 2 (one workgroup per utterance and direction): time-synchronous forward (alpha) and, if
 * beta != NULL, backward (beta) sweeps of gtn.forward_score (ctc.py:50, asg.py:111, stc.py:86,
 * transducer.py:283,287) / viterbi_score.  weights = learnable weight vector indexed by arc_wid
 * (may be NULL).  logz[B] receives the total score.  For WFL_SEMIRING_TROPICAL back-pointers are
 * written to bptr [sum_b T*Q_b] (int32) instead of beta. */
int wfl_lattice_forward(const wfl_lattice_desc* d, const int32_t* ints, const float* floats,
                        const float* xg, int T, const float* weights, int semiring, float* alpha,
                        float* beta, int32_t* bptr, float* logz, void* stream);

/* Stage 3 (all CUs): arc posteriors -> dense emission gradient rows and learnable-weight grads.
 *   dx[b,t,c] (+)= coef[b] * gout * sum_{arcs with label c} gamma_t(arc)
 *   dW[wid]    += coef_w[b] * gout * sum_t gamma_t(arc)           (atomic)
 * (gtn.backward + emissions.grad(): ctc.py:78-81, asg.py:158-168, stc.py:113-116,
 * transducer.py:321-325,333-336).  gout: device scalar (grad_output) or NULL for 1.
 * accumulate != 0 adds into dx instead of overwriting.  If row_lse != NULL the gradient is taken
 * through the fused log-softmax: dx = coef*(post - softmax(x) * sum_c post). */
int wfl_lattice_grad(const wfl_lattice_desc* d, const int32_t* ints, const float* floats,
                     const float* xg, int T, int C, const float* weights, const float* alpha,
                     const float* beta, const float* logz, const float* coef, const float* coef_w,
                     const float* gout, int accumulate, const float* x, const float* row_lse,
                     float* dx, float* dW, void* stream);

/* Stage 2 + the emission gradient in ONE launch (log semiring).  Same sweeps and outputs as
 * wfl_lattice_forward; when the batch allows it (uniform-label acceptors without epsilon arcs:
 * the Transducer's alignment graphs of transducer.py:262-281, CTC-like chains) the launch also
 * holds gradient workgroups that follow the two sweeps outwards from the middle of each utterance
 * and write   dx[b,t,c] = coef[b] * sum_{arcs with label c} gamma_t(arc)      (grad_output = 1)
 * (through the fused log-softmax if row_lse != NULL, as wfl_lattice_grad), and *in_launch is set to 1.
 * Otherwise *in_launch = 0, nothing is written to dx and wfl_lattice_grad does the whole job.
 * After in_launch = 1: scale dx by grad_output if it is not 1 (wfl_scale), then call
 * wfl_lattice_grad_rest, which overwrites the rows of the utterances the launch did not serve
 * (other acceptors; utterances the certificate sent to the log-domain sweeps). */
/* *in_launch on ENTRY: 2 = do not wait for the gradient workgroups before returning to the stream --
 * the caller queues more of its own launches first (the loss reduction) and then calls
 * wfl_lattice_side_join(stream), without which nothing may touch dx, alpha or beta afterwards;
 * any other value: the call joins before it returns. */
int wfl_lattice_side_join(void* stream);
int wfl_lattice_forward_grad(const wfl_lattice_desc* d, const int32_t* ints, const float* floats,
                             const float* xg, int T, int C, const float* weights, float* alpha,
                             float* beta, float* logz, const float* coef, const float* x,
                             const float* row_lse, float* dx, int* in_launch, void* stream);
/* What became of the gradient workgroups beside the sweeps (wfl_lattice_forward_grad), counters of the calling thread's
 * CURRENT DEVICE (a give-up on one GPU does not back off the others):
 *   out[0] calls that launched them            out[1] of those, gates that GAVE UP waiting for the sweeps (kernels of
 *   out[2] gates that went through                     two streams did not run at the same time: a serialising
 *   out[3] calls that took the plain path              profiler, a debugger) -- the call fell back to
 *          during the back-off after a give-up         wfl_lattice_grad_rest for every row, results unaffected
 *   out[4] calls left in the current back-off   out[5] 1 = the environment announces serialised launches
 *   out[6] the gate's bound (polls of ~2 us)           (AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING): never tried
 *   out[7] 1 = the side stream forks from the caller's stream (default)
 * The device counters behind out[1..2] are read without synchronisation: they lag the stream. n <= 8 words. */
int wfl_lattice_diagnostics(uint64_t* out, int n);
int wfl_lattice_grad_rest(const wfl_lattice_desc* d, const int32_t* ints, const float* floats,
                          const float* xg, int T, int C, const float* weights, const float* alpha,
                          const float* beta, const float* logz, const float* coef, const float* gout,
                          const float* x, const float* row_lse, float* dx, void* stream);

/* Tropical back-trace (gtn.viterbi_path, transducer.py:221): path[b, 0..len_b) = caller's arc
 * ids along the best path, in order, including epsilon arcs; path_len[b] its length
 * (<= T + max_levels*(T+1)); path stride = path_stride. */
int wfl_lattice_backtrace(const wfl_lattice_desc* d, const int32_t* ints, const float* floats,
                          const float* alpha, const int32_t* bptr, int T, int32_t* path,
                          int32_t* path_len, int path_stride, void* stream);

/* diagnostic: resident workgroups per CU of the lattice gradient kernel for a dynamic LDS size */
int wfl_debug_grad_occupancy(int lds_bytes);

/* ------------------------------------------------------------------------------------------------
 * Device kernels: dense (fully connected) transitions -- ASG denominator / ASG Viterbi
 *   W [(C+1), C] row-major: W[0,i] start->i, W[1+i, j] = score(prev j -> cur i)  (asg.py:54-69)
 * ------------------------------------------------------------------------------------------------ */
/* Sizes of the scratch buffers: dW_partial [partial_elems] floats (per-workgroup partial sums of the
 * transition gradient) and the opaque workspace `ws` [ws_bytes] shared by forward and grad
 * (per-frame scale bookkeeping of the probability-domain sweeps, per-utterance range flags). */
int wfl_dense_workspace(int B, int T, int C, int64_t* partial_elems, int64_t* ws_bytes);
/* Largest C the dense-transition entry points accept (asg.py:191-209 has no limit: 16384 is an index-width bound).  Up
 * to wfl_dense_on_chip_classes() (192) the (C+1) x C matrix is private to a workgroup -- registers for the
 * probability-domain sweeps, LDS for the log-domain launches behind them; beyond, the frame update of the whole batch
 * runs as one tiled matrix product per frame on the matrix cores, the matrix streamed from L2 (csrc/dense_wide.h): same
 * entry points, same buffers (sizes from wfl_dense_workspace).  wfl_dense_viterbi keeps the matrix in registers up
 * to 256 classes (the max-plus frame has no matrix-core form) and takes a tiled per-frame launch beyond. */
int wfl_dense_max_classes(void);
int wfl_dense_on_chip_classes(void);
/* Byte offset / length of a field of the opaque dense workspace (diagnostics and tests, like wfl_ctc_workspace_field):
 * WFL_DENSE_WS_FLAGS = int32 [B][2], non-zero where a probability-domain sweep (forward, backward) handed the
 * utterance to the log-domain kernels. */
#define WFL_DENSE_WS_FLAGS 0
int wfl_dense_workspace_field(int B, int T, int field, int64_t* offset_bytes, int64_t* length_bytes);
/* forward_score(intersect(emissions, transitions)) (asg.py:114): logz [B]; alpha, beta [B,T,C] are
 * opaque inputs of wfl_dense_grad in the log semiring (scaled probabilities for utterances served
 * by the probability-domain sweep, log scores for utterances it had to hand to the log-domain
 * sweep; ws records which), max-plus scores in the tropical semiring (ws may be NULL there). */
int wfl_dense_forward(const float* x, const float* W, int B, int T, int C, int semiring,
                      float* alpha, float* beta, int32_t* bptr, float* logz, void* ws, void* stream);
/* dx[b,t,i] = (accumulate ? dx : 0) + gout*addend[b,t,i] + coef[b]*gout*post_t(i);
 * dW       = (accumulate ? dW : 0) + gout*dW_addend    + sum_b coef_w[b]*gout*transition posteriors.
 * `addend` [B,T,C] and `dW_addend` [(C+1),C] (either may be NULL): terms computed for gout = 1 before gout was
 * known -- the ASG numerator's posteriors, which the criterion computes during forward on a second stream under the
 * (longer) denominator sweeps (asg.py:158-168 adds the two gradients in backward). */
int wfl_dense_grad(const float* x, const float* W, int B, int T, int C, const float* alpha,
                   const float* beta, const float* logz, const float* coef, const float* coef_w,
                   const float* gout, int accumulate, const float* addend, const float* dW_addend,
                   float* dx, float* dW, float* dW_partial, const void* ws, void* stream);
/* The same two calls in parts, for a caller that runs them on two streams (criterions/asg.py through
 * csrc/torch_ops.cpp::asg_forward): the probability-domain launches serve every utterance whose transition matrix has a
 * bounded dynamic range, the log-domain launches behind them only what those flagged (normally nothing: they return at
 * once) -- different utterances, disjoint rows of every output.  A caller may put WFL_DENSE_REPAIR on a second stream,
 * ordered after the WFL_DENSE_MAIN part of wfl_dense_forward_parts (it reads the flags that part writes), and run
 * WFL_DENSE_REDUCE (the sum of the per-workgroup transition-gradient partials into dW) once both gradient parts are done.
 * WFL_DENSE_ALL on one stream is wfl_dense_forward / wfl_dense_grad.  Beyond wfl_dense_on_chip_classes() and in the
 * tropical semiring the work is one piece: it goes with WFL_DENSE_MAIN, the other parts do nothing. */
#define WFL_DENSE_MAIN 1
#define WFL_DENSE_REPAIR 2
#define WFL_DENSE_REDUCE 4
#define WFL_DENSE_ALL 7
int wfl_dense_forward_parts(const float* x, const float* W, int B, int T, int C, int semiring,
                            float* alpha, float* beta, int32_t* bptr, float* logz, void* ws, int parts, void* stream);
int wfl_dense_grad_parts(const float* x, const float* W, int B, int T, int C, const float* alpha,
                         const float* beta, const float* logz, const float* coef, const float* coef_w,
                         const float* gout, int accumulate, const float* addend, const float* dW_addend,
                         float* dx, float* dW, float* dW_partial, const void* ws, int parts, void* stream);
/* viterbi_path(intersect(emissions, transitions)).labels_to_list() (asg.py:225-226):
 * path [B,T] int32 emission labels.  Ties: lowest previous label, then lowest final label.
 * alpha [B,T,C]: the max-plus vectors (an output).  bptr: unused since round 5 (neither read nor written, may be NULL):
 * the back-trace re-derives the one back-pointer per frame it follows from the stored vectors; wfl_dense_forward in the
 * tropical semiring still fills a back-pointer buffer. */
int wfl_dense_viterbi(const float* x, const float* W, int B, int T, int C, float* alpha,
                      int32_t* bptr, int32_t* path, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device kernels: ConvTransduce1D (transducer.py:351-556)
 *   out[b, w, k] = forward_score | viterbi_score (intersect(x[b, w*stride : w*stride+ks, :],
 *                  make_kernel_graph(lexicon[k])))                       transducer.py:485-500
 *   x [B,T,C] already padded by the caller (T >= ks, transducer.py:468-470), Tout = (T-ks)/stride+1.
 *   ktab [K][36] int32 describes the lexicon: {L, skip mask (bit i: arc 2i-1 -> 2i+1 exists),
 *   tok[16], base[16] (index in kernel_params of entry k's arc 2i -> 2i+1), index of its arc 0->0,
 *   pad}; arc ids follow the insertion order of make_kernel_graph (transducer.py:351-364).
 *   params: kernel_params (NULL: all arc weights 0).  L <= 15, ks <= 16.
 * ------------------------------------------------------------------------------------------------ */
#define WFL_CONV_SPIKE 1          /* flags: no self loops on sub-token states */
#define WFL_CONV_BLANK_OPTIONAL 2 /* flags: state 2L-1 accepts, skip arcs between different sub-tokens */
int wfl_conv_forward(const float* x, int B, int T, int C, const int32_t* ktab, int K, int ks,
                     int stride, int blank, int flags, const float* params, int semiring, float* out,
                     void* stream);
/* dx [B,T,C] = sum_{w,k} delta[b,w,k] * d out[b,w,k] / d x (overwritten); dparams [num_arcs]
 * (accumulated, may be NULL) likewise for kernel_params (transducer.py:514-552). */
int wfl_conv_grad(const float* x, int B, int T, int C, const int32_t* ktab, int K, int ks, int stride,
                  int blank, int flags, const float* params, int semiring, const float* delta,
                  float* dx, float* dparams, void* stream);

/* STC's alphabet augmentation (stc.py:199-220) in one launch each way.  x: log-probabilities [T, B, C] (the module's
 * input layout); select[K]: the classes of the batch, the blank (0) first, each once; inv[C]: select's inverse (-1 for a
 * class that is not selected).  out [B, T, 2K] = (selected columns | <star> = logsumexp over c >= 1 | <star>\token for
 * select[1..]), lse [T * B] the rows' <star> (kept for the backward); dx [T, B, C] is OVERWRITTEN with the gradient for
 * the upstream g [B, T, 2K]. */
int wfl_stc_augment(const float* x, int T, int B, int C, const int32_t* select, int K, float* out, float* lse, void* stream);
int wfl_stc_augment_grad(const float* x, int T, int B, int C, const int32_t* select, const int32_t* inv, int K,
                         const float* lse, const float* g, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device kernels: CTC fast path (create_ctc_graph + intersect + forward_score + backward of
 * ctc.py:15-94; banded recursion with register-resident state, no lattice arrays).
 *   targets: device int32 flat, offsets: device int64 [B+1]; requires max target length <= 255 and
 *   C <= 16384 (targets of more than 63 labels: C <= 602) -- the gradient tiles live in LDS
 *   (up to 63: one position per lane; longer: two to four positions per lane)
 *   (longer targets: WFL_ERR_UNSUPPORTED -> use wfl_lattice_pack_ctc + the lattice engine).
 *   loss_b = -logZ_b is written to nll[B]; dx = coef[b]*gout*posteriors (dense rows).
 *   Base-2 log-domain arithmetic, block-renormalised; the chain stores one checkpoint per 16
 *   frames and the gradient kernel recomputes inside the blocks (csrc/ctc_kernels.hip).
 * ------------------------------------------------------------------------------------------------ */
int wfl_ctc_workspace(int B, int T, int C, int max_len, int64_t* ws_elems);
/* Diagnostics: offset (in floats) and length of a bookkeeping field inside the CTC workspace.
 *   WFL_CTC_WS_REJECTED  int32[B]    (unused since round 4: the three-launch lane-exponent step is retired; zeros)
 *   WFL_CTC_WS_STATUS    int32[2]    pipelined step: [0] a gradient wave gave up waiting, [1] utterances repaired
 *   WFL_CTC_WS_LOG2Z     double[B]   log2 Z per utterance
 *   WFL_CTC_WS_ZRANGE    int64[B][2] pipelined lane-exponent step: min / max over the blocks of log2 Z * 2^16
 *   WFL_CTC_WS_DEBUG     per-wave cycle counters of a -DWFL_MITM_STATS=1 build (length 0 in a normal build)
 *   WFL_CTC_WS_CLOCK     int64[B][2][2] meet-in-the-middle step: the device's constant 100 MHz clock at the entry of every sweep's
 *                        workgroup and at the exit of its last wave (the launch's own duration, measured on the device) */
#define WFL_CTC_WS_REJECTED 0
#define WFL_CTC_WS_STATUS 1
#define WFL_CTC_WS_LOG2Z 2
#define WFL_CTC_WS_ZRANGE 3
#define WFL_CTC_WS_DEBUG 4
#define WFL_CTC_WS_CLOCK 5
int wfl_ctc_workspace_field(int B, int T, int max_len, int field, int64_t* offset_elems, int64_t* length_elems);
/* alpha and beta chains: writes nll[B] = -log Z_b and the 16-frame checkpoints into ws.
 * flags must be 0 (the log-domain chain; the lane-exponent variant of this call was retired in round 4 --
 * wfl_ctc_forward_backward is the training step). */
int wfl_ctc_forward(const float* x, int B, int T, int C, const int32_t* targets,
                    const int64_t* offsets, int max_len, int blank, int flags, float* ws, float* nll,
                    void* stream);
/* Per-call extras of wfl_ctc_forward_backward_call -- everything the step remembers or assumes beyond its arguments
 * is the CALLER's:
 *   n_labels    the number of labels behind `targets` the caller vouches for (= offsets[B] as the host knows it; 0:
 *               unknown).  With it a workgroup asks for its labels together with its offsets (one memory round trip at
 *               kernel entry instead of two) under the guess that every target has max_len labels -- true for bucketed
 *               batches and the benchmark's; a ragged batch asks again, results are the same.
 *   host_state  2 x int32 of PINNED host memory (hipHostMalloc / a pin_memory tensor), zeroed by the caller once, one per
 *               criterion / workspace and never shared between concurrent calls -- or NULL.  [0] is written by the
 *               repair launch (how many utterances of the last lane-exponent step it recomputed; system scope, nobody
 *               waits for it), [1] is the step's own counter.  When the LAST lane-exponent step recomputed more than an
 *               eighth of its utterances the call goes straight to the log-domain step, and every 16th such call tries
 *               the lane-exponent step again.  NULL: every call starts with the lane-exponent step.  Results are within
 *               the parity bar on either path (bit-identical only on the same path); zero the words to forget. */
typedef struct wfl_ctc_call {
  int64_t n_labels;
  int32_t* host_state;
} wfl_ctc_call;

/* wfl_ctc_forward and wfl_ctc_grad as ONE pipelined launch: gradient waves wait for the checkpoints
 * they need and run while the chains are still sweeping.  Same outputs (nll, dx); posteriors are
 * normalised per 16-frame block by the Z the block reproduces.
 * Targets of up to 63 labels: chains and gradient blocks run in lane-exponent
 * (probability-domain) arithmetic; every block certifies its result (log2 Z reproduced to 1.5e-4, the
 * posteriors of its frames sum to one to 2e-4) and a second, normally empty launch recomputes rejected
 * utterances in the log domain -- the caller always receives certified or log-domain results.
 * WFL_CTC_PIPELINE=log in the environment selects the log-domain launch throughout.  If loss_out is
 * not NULL it also receives mean_b(loss_scale[b] * nll[b]) (ctc.py:68-69; loss_scale NULL = 1),
 * reduced in a fixed order by the last chain to finish -- no separate wfl_reduce_loss launch.
 * If row_lse is not NULL, x holds RAW scores and row_lse[b*T + t] their log-sum-exp over the classes
 * (wfl_row_lse): the log_softmax of the CTC module (ctc.py:107) is fused into the launch, forward
 * and backward (dx is then the gradient w.r.t. the raw scores). */
int wfl_ctc_forward_backward(const float* x, int B, int T, int C, const int32_t* targets,
                             const int64_t* offsets, int max_len, int blank, float* ws, float* nll,
                             const float* coef, const float* gout, float* dx, const float* loss_scale,
                             float* loss_out, const float* row_lse, void* stream);
/* wfl_ctc_forward_backward with the per-call extras above (call may be NULL: the same as the function above, which
 * keeps no memory between calls) */
int wfl_ctc_forward_backward_call(const float* x, int B, int T, int C, const int32_t* targets,
                                  const int64_t* offsets, int max_len, int blank, float* ws, float* nll,
                                  const float* coef, const float* gout, float* dx, const float* loss_scale,
                                  float* loss_out, const float* row_lse, const wfl_ctc_call* call, void* stream);
/* out[r] = logsumexp_c x[r*C + c] for r < rows (NaN counts as -inf) */
int wfl_row_lse(const float* x, int64_t rows, int C, float* out, void* stream);
/* out[r] = the first c with x[r*C + c] == max_c x[r*C + c] (NaN counts as -inf; a row without a finite score: 0) --
 * viterbi_path of the bare emissions graph (transducer.py:205-216 without transitions), one frame per row */
int wfl_row_argmax(const float* x, int64_t rows, int C, int32_t* out, void* stream);
/* dense gradient rows, recomputed block by block from the checkpoints of wfl_ctc_forward */
int wfl_ctc_grad(const float* x, int B, int T, int C, const int32_t* targets, const int64_t* offsets,
                 int max_len, int blank, const float* ws, const float* nll, const float* coef,
                 const float* gout, float* dx, void* stream);

/* small device utilities used by the Python layer (kept here so the product never needs a
 * torch op inside the timed path) */
/* dst[0..nbytes) (device) = src_pinned[0..nbytes) (PINNED host memory, e.g. hipHostMalloc / a pin_memory tensor), by
 * a kernel that reads the host buffer through its device-visible address; both 16-byte aligned.  For the small
 * per-batch uploads of the criteria (targets, packed lattices): unlike hipMemcpyAsync, which on this stack sometimes
 * blocks the calling thread until the stream has drained (measured: 7 us .. 5 ms for 23 KB behind queued kernels),
 * a launch is always asynchronous.  The host buffer must stay untouched until the stream has passed the launch. */
int wfl_upload(void* dst, const void* src_pinned, int64_t nbytes, void* stream);
/* v[0..n) *= s[0] on the device; a no-op pass when s[0] == 1 (upstream gradient of a scalar loss) */
int wfl_scale(float* v, int64_t n, const float* s, void* stream);
/* out[0] = (1/B) * sum_b sign * scale[b] * (vals[b] - minus[b])  (+ out[0] if accumulate); minus may be NULL (0).
 * With `minus` the two-term criteria (ASG: denominator - numerator, asg.py:116-121; Transducer) reduce in one launch. */
int wfl_reduce_loss(const float* vals, const float* minus, const float* scale, int B, float sign,
                    int accumulate, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WFL_H_ */
