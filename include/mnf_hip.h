/*
 * mnf_hip.h -- C ABI of libmnf_hip.so: the coupling-flow hot path of janosh/torch-mnf
 * as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference is pure Python on PyTorch and has no FFI of its own; the interface each
 * entry point replaces is the duck-typed Flow module method it computes
 * (paths below /root/reference):
 *
 *   mnf_affine_half      AffineHalfFlow.forward / .inverse      torch_mnf/flows/affine_half_flow.py:44-66
 *   mnf_nsf_cl           NSF_CL.forward / .inverse              torch_mnf/flows/spline_flow.py:249-285
 *   mnf_nsf_ar           NSF_AR.forward / .inverse              torch_mnf/flows/spline_flow.py:201-235
 *   mnf_rqs              unconstrained_RQS                      torch_mnf/flows/spline_flow.py:29-68
 *   mnf_rnvp             RNVP.forward                           torch_mnf/flows/rnvp.py:25-39
 *   mnf_maf              MAF / IAF .forward / .inverse          torch_mnf/flows/maf.py:39-72
 *   mnf_affine_const     AffineConstantFlow.forward / .inverse  torch_mnf/flows/affine_constant_flow.py:18-26
 *   mnf_linear_rows      Glow.forward / .inverse (x @ W)        torch_mnf/flows/glow.py:26-37
 *   mnf_glow_weight      Glow._assemble_W, W^-1, log_det        torch_mnf/flows/glow.py:20-37
 *   mnf_glow_actnorm_inv Glow.inverse + ActNormFlow.inverse     torch_mnf/flows/glow.py:33-37,
 *     (+ _logprob)       (a block's pair on the way x -> z;     torch_mnf/flows/affine_constant_flow.py:22-26,
 *                        closing a density pass: + log_prob)    torch_mnf/flows/core.py:46-49
 *   mnf_gauss_logprob(_sq) base.log_prob + the callers' mean    torch_mnf/flows/core.py:46-49,
 *                                                               examples/half_moons.ipynb:183-186
 *   mnf_sample_z0        MNFLinear.sample_z prologue            torch_mnf/layers/mnf_linear.py:58-62
 *   mnf_mnf_linear_fwd   MNFLinear.forward behind sample_z      torch_mnf/layers/mnf_linear.py:46-56
 *   mnf_mnf_kl_fwd       MNFLinear.kl_div / MNFConv2d.kl_div    torch_mnf/layers/mnf_linear.py:66-90,
 *                        behind their two flows                 torch_mnf/layers/mnf_conv.py:100-133
 *   log_det accumulation NormalizingFlow.forward / .inverse     torch_mnf/flows/core.py:17-35
 *                        (the `accumulate` flag of every layer entry point: log_det += ld)
 *
 * Contract:
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller owns all
 *     buffers; the library allocates nothing and keeps no state between calls;
 *   - all tensors are fp32, row-major contiguous: x, y are (rows, dim); log_det is (rows,);
 *   - y must not alias x (layers are out-of-place, as in the reference);
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*); no host
 *     synchronisation inside, so calls can be captured into a hipGraph;
 *   - return value: MNF_OK or a negative MNF_ERR_* code; nothing is thrown across the ABI;
 *   - re-entrant and thread-safe.
 *
 * Parameter layouts
 *   "flat": the module's state_dict tensors concatenated in state_dict order, each in its
 *   own row-major layout (nn.Linear weight is (out, in)).  For AffineHalfFlow: s_net then
 *   t_net (an absent net contributes nothing); NSF_CL: f1 then f2; RNVP: net, t, s.
 *   "image": the same numbers rearranged into per-lane MFMA A-operand order for the
 *   specialised kernels; built on the device by mnf_pack_gather from an index table that
 *   mnf_*_image_index fills on the host once per configuration.
 */
#ifndef MNF_HIP_H
#define MNF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNF_OK 0
#define MNF_ERR_INVALID_ARG (-1) /* null pointer, odd dim, negative size, ...                     */
#define MNF_ERR_UNSUPPORTED (-2) /* shape outside what the kernels cover (e.g. layer too wide)    */
#define MNF_ERR_LAUNCH (-3)      /* hipLaunchKernel reported an error; see mnf_last_hip_error()   */
#define MNF_ERR_NO_DEVICE (-4)   /* no gfx950 device visible                                      */
#define MNF_ERR_DOMAIN (-5)      /* spline: min bin width/height * K > 1 (ValueError in reference) */

#define MNF_MAX_LINEAR 8 /* Linear layers per conditioner net (hidden layers + 1) */

/* Bumped whenever an entry point's signature or meaning changes; mnf_abi_version() returns the value the
 * library was built with, so a binding can refuse a stale build. */
#define MNF_ABI_VERSION 14
int mnf_abi_version(void);
const char* mnf_error_string(int code);
/* hipError_t of the last failed launch on the calling thread (0 if none). */
int mnf_last_hip_error(void);
/* Which kernel family the process's most recent layer call ran (any thread: autograd launches gradients on its own): a static string such as "ahf_split_stack",
 * "nsf_mfma_split", "nsf_bwd_tile", "rnvp_resident" -- or "ahf_generic", "nsf_generic", "nsf_bwd_generic", ... for the
 * any-shape kernels, which are 20-40 x slower at large batches.  The reference has no counterpart (one code path); here a
 * caller can see which side of a shape cliff a layer landed on (shape -> kernel table).  "" before the
 * first launch. */
const char* mnf_last_kernel(void);
/* 1 when MNF_DETERMINISTIC is set in the environment (read once per process): the RNVP and MNFLinear gradient launches
 * (mnf_rnvp_bwd_mfma, mnf_mnf_linear_bwd) then form their parameter sums without float atomics -- one block per row part
 * in an extension of the caller's workspace (their *_workspace_bytes queries include it), added up in a fixed order -- so
 * that a training step repeats bit for bit, as the reference's does under torch.manual_seed (tests/test_flows.py:11).
 * The split AffineHalfFlow and the NSF_CL tile gradient launches reduce in a fixed order in every mode; the [Glow, ActNorm]
 * pair, the fp32-MFMA AffineHalfFlow kernel and sample_z have *_det entry points of their own.  The fp32 fix-up passes
 * (tiles / row groups whose values left the split range, handed back as a list in any order) then sort their list on the
 * device -- in place, also where this header passes it as const -- and run as ONE workgroup, so that their sums follow
 * the list (slow; such rows are the exception).  tools/soak_determinism_train.py: 0 differing tensors with a tenth of
 * the rows forced onto those passes.  The run-time-shaped gradient kernels (mnf_*_bwd_rt) return MNF_ERR_UNSUPPORTED
 * under the switch: their shapes then take the VALU gradient kernels, which add atomically across workgroups -- the
 * switch covers the shapes with per-shape gradient kernels (the reference's configurations), not every shape. */
int mnf_deterministic(void);
/* Number of visible devices whose gcnArchName starts with gfx950 (0 = none / no driver). */
int mnf_device_count(void);

/* ---------------------------------------------------------------- AffineHalfFlow */
/* hidden: n_hidden sizes of the hidden layers (reference default {24,24,24}).
 * Three kernels, fastest first:
 *   split   image and split_image given, shape supported: conditioner nets on the f16 matrix pipe in
 *           split (hi + lo) fp32 arithmetic -- three f16 MFMAs per fp32 product, fp32 accumulation,
 *           ~22 significant bits per product; tiles whose operands leave the f16 range are recomputed
 *           on the fp32 MFMA path inside the same launch (needs `image`)
 *   fp32    image given (mnf_affine_half_image_floats() floats), split_image NULL: fp32 MFMAs
 *   generic otherwise: reads `flat`, any shape
 * Shapes with operand images (mnf_affine_half_image_floats() > 0): three hidden layers of at most 32 units each
 * (run at 16 / 24 / 32 units with structural zeros), any even dim <= 256 (a half narrower than its 16/32/64/128-
 * column tile is zero-padded in the image and runs the stack kernel's ragged variant with one layer; hidden width 32
 * up to dim 128; hidden widths 33 .. 64 run at 64 units at dim = 32, 64 and 128, one layer per launch), has_scale / has_shift (an absent net is an all-zero operand set: s = 0 or t = 0 exactly).
 * log_det may be NULL (not computed). accumulate != 0: log_det += ld.
 * force_generic == 1 selects the VALU any-shape kernel, == 2 the run-time-shaped matrix-core kernel (mnf_ahf_rt.hip:
 * any h_sizes of length >= 1 with widths 4 .. 256, any even dim; weights converted from `flat` inside the kernel) whatever
 * the shape's specialised kernels (tests and tools/coverage_map.py compare the three).  With force_generic == 0 a shape
 * without a specialised kernel runs the run-time-shaped one from 2,048 rows on and the VALU one below. */
int mnf_affine_half(const float* x, float* y, float* log_det, int accumulate,
                    const float* flat, const float* image, const void* split_image,
                    int64_t rows, int dim, int parity, int inverse,
                    int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                    int force_generic, void* stream);
/* Same layer, additionally writing y_sqnorm[r] = |y_r|^2 (rows,) so that the base log-prob of
 * a standard-normal base needs no second pass over y.  Only the
 * specialised kernel produces it: MNF_ERR_UNSUPPORTED otherwise (callers then run
 * mnf_gauss_logprob on y). */
int mnf_affine_half_sq(const float* x, float* y, float* log_det, float* y_sqnorm, int accumulate,
                       const float* flat, const float* image, const void* split_image,
                       int64_t rows, int dim, int parity, int inverse,
                       int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                       int force_generic, void* stream);
/* Whole-stack fusion: n_layers AffineHalfFlow layers of one shape in ONE launch, rows kept
 * in registers across layers.  images = the layers' operand
 * images back to back (layer 0 first); split_images = their split images back to back, or NULL for
 * the fp32 MFMA kernel; layers are applied 0..L-1 (forward) or L-1..0 (inverse).
 * intermediates: NULL, or a (n_layers - 1, rows, dim) buffer that receives the output of every layer but
 * the last in application order -- each intermediate tensor is then written once and never re-read
 * (4*dim*(n_layers+1) + 8 bytes per row for the stack instead of (8*dim+8)*n_layers), which is how
 * NormalizingFlow.forward/inverse run a chain of equal AffineHalfFlow layers while still returning all
 * of them.  HBM traffic with intermediates == NULL: 8*dim + 8 bytes per row for the whole stack.
 * log_prob (rows,) / log_prob_sum (device double, ADDED to; caller zeroes it), both optional (split kernel only,
 * need log_det): the standard-normal base log-prob epilogue log_det - |y|^2/2 - dim/2 log(2 pi) of
 * mnf_gauss_logprob fused into the same launch (when the stack is the whole inverse pass of a model).
 * MNF_ERR_UNSUPPORTED for shapes without a fused kernel: the split kernel covers every shape that has a split image
 * except hidden width 32 at full-tile dim <= 64 with n_layers > 1 (no faster than one launch per layer); the fp32
 * kernel behind it dim in {32, 64} with hidden width 24 or 16. */
int mnf_affine_half_stack(const float* x, float* y, float* intermediates, float* log_det, float* y_sqnorm,
                          float* log_prob, double* log_prob_sum, int accumulate,
                          const float* images, const void* split_images, const int* parity_host, int n_layers,
                          int64_t rows, int dim, int inverse,
                          int n_hidden, const int* hidden_host, void* stream);
/* Split image of one layer: n_split_words 32-bit words of packed f16 (hi | scaled lo) operands, then
 * n_plain_words fp32 words (biases), then MNF_SPLIT_TAIL_WORDS words written by the pack call (word 0
 * = bit pattern of max |weight|: the kernels take the fp32 path when it leaves the f16 range).
 * _split_index fills 2 * n_split_words + n_plain_words entries for mnf_pack_gather_split.
 * MNF_ERR_UNSUPPORTED when the configuration has no split kernel. */
#define MNF_SPLIT_TAIL_WORDS 4
int mnf_affine_half_split_layout(int dim, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                                 int64_t* n_split_words, int64_t* n_plain_words);
int mnf_affine_half_split_index(int dim, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                                int32_t* idx_host);
/* 0 when the configuration has no specialised (MFMA) kernel. */
int64_t mnf_affine_half_image_floats(int dim, int n_hidden, const int* hidden_host,
                                     int has_scale, int has_shift);
/* idx_host[i] = index into `flat` feeding image[i], or -1 for a structural zero. */
int mnf_affine_half_image_index(int dim, int n_hidden, const int* hidden_host,
                                int has_scale, int has_shift, int32_t* idx_host);
int64_t mnf_affine_half_flat_floats(int dim, int n_hidden, const int* hidden_host,
                                    int has_scale, int has_shift);

/* image[i] = flat[idx[i]] for idx[i] >= 0, 0 for -1, the constant 80.0 for -2 (the scale bias of an RNVP output
 * dim that only exists as padding: gate 1, log gate 0)  (device-side repack after a weight update). */
/* Split-image pack (device): split word w = two f16 halves, half h from entry e = idx[2 w + h]:
 * e < 0 -> 0; else v = flat[e & 0x3fffffff], hi = f16(v), and the half is hi, or f16((v - hi) * 2^11)
 * when bit 30 of e is set.  Plain word w = flat[idx[2 n_split_words + w]] (or 0).  Also writes the
 * MNF_SPLIT_TAIL_WORDS tail.  image holds n_split_words + n_plain_words + MNF_SPLIT_TAIL_WORDS words. */
int mnf_pack_gather_split(const float* flat, const int32_t* idx_dev, void* image, int64_t n_split_words,
                          int64_t n_plain_words, void* stream);
int mnf_pack_gather(const float* flat, const int32_t* idx, float* image, int64_t n, void* stream);
/* The same for n_images layers of ONE shape in one launch each: image k (back to back in `images`) is built from
 * flat + k * flat_stride with the shared index table.  A run of equal AffineHalfFlow layers repacks all its
 * operand images after a weight update with one parameter concatenation and two launches (a training step
 * otherwise spends more time launching per-layer packs than computing). */
int mnf_pack_gather_batch(const float* flat, const int32_t* idx, float* images, int64_t n, int n_images,
                          int64_t flat_stride, void* stream);
int mnf_pack_gather_split_batch(const float* flat, const int32_t* idx_dev, void* images, int64_t n_split_words,
                                int64_t n_plain_words, int n_images, int64_t flat_stride, void* stream);

/* ------------------------------------------------------------------------ NSF_CL */
/* f1, f2 = MLP(dim/2, n_h, n_h, n_h, (3K-1)*dim/2); `hidden` generalises (n_h,n_h,n_h).
 * MFMA kernels (image given): dim 32 or 64, K 8 or 5, three hidden layers of at most 16 units (run at 8 or 16 with
 * structural zeros; at dim 64 with K = 5 at most 8); the generic kernel (flat) otherwise. */
int mnf_nsf_cl(const float* x, float* y, float* log_det, int accumulate,
               const float* flat, const float* image, const void* split_image,
               int64_t rows, int dim, int K, float tail_bound, int inverse,
               int n_hidden, const int* hidden_host, int force_generic, void* stream);
/* Opt-in fusion of the reference's [ActNorm, Glow, NSF_CL] block into one kernel (SURVEY.md 8f
 * rank 3; drops the block's two intermediate tensors, so it is not what NormalizingFlow.forward
 * returns by default).  ActNorm and Glow collapse to  row @ A + b  (A: dim x dim, given as the
 * MFMA operand image of mnf_linear_rows_image_index followed by the dim bias values) and the
 * row-independent log-det constant ld_const:
 *   forward:  y = NSF_CL(x @ A + b)            A = diag(e^s) W,      b = t @ W
 *   inverse:  y = NSF_CL^-1(x) @ A + b         A = W^-1 diag(e^-s),  b = -t e^-s
 * mid1, mid2 (both NULL, or (rows, dim) buffers): the block's two intermediate tensors in application order,
 * written once from registers and never re-read -- forward ActNorm(x) and Glow(ActNorm(x)), inverse
 * NSF_CL^-1(x) and Glow^-1 of it; this is how NormalizingFlow runs the block as one launch while still
 * returning every tensor.  They need scale_shift = [exp(s) (dim), t (dim)] of the ActNorm layer.
 * log_prob (rows,) / log_prob_sum (device double, ADDED to; caller zeroes it), both optional (need log_det): the
 * standard-normal base log-prob epilogue log_det - |y|^2/2 - dim/2 log(2 pi) and its fp64 sum over rows in the
 * same launch (when the block is the last launch of a density pass), as in mnf_affine_half_stack.
 * MNF_ERR_UNSUPPORTED when the shape has no fused kernel (callers run the three layers). */
int mnf_nsf_cl_fused(const float* x, float* y, float* log_det, int accumulate, const float* image,
                     const void* split_image, const float* aff, float ld_const, const float* scale_shift,
                     float* mid1, float* mid2, float* log_prob, double* log_prob_sum,
                     int64_t rows, int dim, int K, float tail_bound,
                     int inverse, int n_hidden, const int* hidden_host, void* stream);
int64_t mnf_nsf_cl_flat_floats(int dim, int K, int n_hidden, const int* hidden_host);
int64_t mnf_nsf_cl_image_floats(int dim, int K, int n_hidden, const int* hidden_host);
int mnf_nsf_cl_image_index(int dim, int K, int n_hidden, const int* hidden_host, int32_t* idx_host);
/* Split image (see mnf_affine_half_split_layout): with image AND split_image the conditioner nets run on
 * v_mfma_f32_16x16x16_f16 in split fp32 arithmetic; a half step whose operands leave the f16 range is
 * redone with fp32 MFMAs inside the same launch. */
int mnf_nsf_cl_split_layout(int dim, int K, int n_hidden, const int* hidden_host, int64_t* n_split_words,
                            int64_t* n_plain_words);
int mnf_nsf_cl_split_index(int dim, int K, int n_hidden, const int* hidden_host, int32_t* idx_host);

/* Elementwise unconstrained rational-quadratic spline: inputs (n,), W,H (n,K), D (n,K-1). */
int mnf_rqs(const float* inputs, const float* W, const float* H, const float* D,
            float* outputs, float* logabsdet, int64_t n, int K, float tail_bound,
            int inverse, void* stream);

/* -------------------------------------------------------------- RNVP (masked/gated) */
/* mask: (rows, dim) floats in {0,1}, supplied by the caller (the reference draws
 * torch.bernoulli per call, rnvp.py:28).  net = MLP(dim, hidden...); t, s = Linear(h_last, dim).
 * MFMA kernels (image / split_image given): one hidden layer of at most 64 units (run at 30, 50 or 64 with structural
 * zeros), any dim >= 49 (dim % 16 != 0: zero-padded operand images, masked row accesses; the padded dims' scale bias
 * is the constant 80 so that they add nothing to log_det); the generic kernel (flat) otherwise. */
int mnf_rnvp(const float* z, const float* mask, float* x, float* log_det, int accumulate,
             const float* flat, const float* image, const void* split_image,
             int64_t rows, int dim, int n_hidden, const int* hidden_host,
             int force_generic, void* stream);
/* Same layer with the mask generated inside the kernel when mask == NULL: element (r, j) is bit
 * (j & 31) of a 32-bit hash of (seed, r, j >> 5) -- a stateless Bernoulli(0.5) stream owned by
 * this library.  mnf_rnvp_mask writes exactly that mask as floats, so a
 * seeded call can be reproduced with an explicit mask (and by the CPU oracle). */
int mnf_rnvp_seeded(const float* z, const float* mask, uint64_t seed, float* x, float* log_det,
                    int accumulate, const float* flat, const float* image, const void* split_image,
                    int64_t rows, int dim, int n_hidden, const int* hidden_host,
                    int force_generic, void* stream);
int mnf_rnvp_mask(uint64_t seed, float* mask, int64_t rows, int dim, void* stream);
/* The training form of mnf_rnvp_seeded: when y_out != NULL and the kernel that takes the call can (the split-arithmetic
 * kernels: the register-resident one and the streaming one, either mask form), it also writes y = net(mask * z) -- (rows, mnf_rnvp_y_floats_per_row(...)) floats,
 * NaN for rows it recomputed in fp32 -- for mnf_rnvp_bwd_mfma_phases, and sets *y_written_host = 1; otherwise
 * *y_written_host = 0 and y_out is untouched.  Everything else as mnf_rnvp_seeded. */
int mnf_rnvp_y_floats_per_row(int n_hidden, const int* hidden_host);
int mnf_rnvp_seeded_train(const float* z, const float* mask, uint64_t seed, float* x, float* log_det, int accumulate,
                          const float* flat, const float* image, const void* split_image, int64_t rows, int dim,
                          int n_hidden, const int* hidden_host, int force_generic, float* y_out, int* y_written_host,
                          void* stream);
/* Few rows through a layer with ONE hidden layer of at most 64 units take latency kernels (mnf_rnvp_few.hip) instead of
 * the streaming ones when `flat` is given and force_generic is 0: weight rows read coalesced by a wave per dim, DPP
 * wave sums, no operand image.
 *   mnf_rnvp_seeded   a workgroup per two rows: up to MNF_RNVP_FEW_FWD_ROWS rows with an explicit mask (the reference's
 *                     batch of 128 rows of 800 dims: 103 -> 35 us), up to 64 with the in-kernel mask (the register-
 *                     resident kernels are as fast from ~100 rows); one row of 800 dims: 86 -> 22 us
 *   mnf_rnvp_bwd      up to MNF_RNVP_FEW_ROWS rows -- what the MNF layers' kl_div runs (one row per call,
 *                     mnf_linear.py:67/84, mnf_conv.py:90-101/127): ONE workgroup owns every parameter gradient, no
 *                     atomics (one row of 800 dims: 150 -> 49 us)
 * MNF_RNVP_FEW=0 in the environment switches both off. */
#define MNF_RNVP_FEW_ROWS 2
#define MNF_RNVP_FEW_FWD_ROWS 512
#define MNF_RNVP_FEW_BWD_ROWS 512
/* The gradient kernel of the same family as a grid for batch-sized calls (3 .. MNF_RNVP_FEW_BWD_ROWS rows; the reference
 * trains at 128): a workgroup per two rows WRITES its copy of the parameter gradients to `workspace`
 * (mnf_rnvp_bwd_few_workspace_floats(...) floats; 0 = shape or row count not covered) and a second launch adds the
 * copies to grad_flat (ADDED to, as in mnf_rnvp_bwd; grad_z written).  128 rows of 800 dims: ~85 us against ~190 for the
 * three matrix-core launches, whose first is one workgroup's two sweeps over all dims at that size. */
int64_t mnf_rnvp_bwd_few_workspace_floats(int64_t rows, int dim, int n_hidden, const int* hidden_host);
int mnf_rnvp_bwd_few(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                     float* grad_z, float* grad_flat, const float* flat, float* workspace, int64_t rows, int dim,
                     int n_hidden, const int* hidden_host, void* stream);
/* 1 when mnf_rnvp_seeded takes that kernel for this shape (the caller then needs no operand image for the forward call). */
int mnf_rnvp_few_rows_ok(int64_t rows, int dim, int n_hidden, const int* hidden_host, int explicit_mask);
/* MNFLinear.sample_z's prologue fused into its first flow (torch_mnf/layers/mnf_linear.py:58-64):
 * x = RNVP(q0_mean + sqrt(exp(q0_log_var)) * eps); z0 is formed in the kernel's loads and never stored.
 * mask == NULL: in-kernel mask from `seed`.  Needs image and split_image (split MFMA kernel) and dim <= 1024;
 * MNF_ERR_UNSUPPORTED otherwise (callers then run mnf_sample_z0 followed by mnf_rnvp_seeded). */
int mnf_rnvp_sample(const float* eps, const float* q0_mean, const float* q0_log_var, const float* mask,
                    uint64_t seed, float* x, float* log_det, int accumulate, const float* image,
                    const void* split_image, int64_t rows, int dim, int n_hidden, const int* hidden_host,
                    void* stream);
int64_t mnf_rnvp_flat_floats(int dim, int n_hidden, const int* hidden_host);
int64_t mnf_rnvp_image_floats(int dim, int n_hidden, const int* hidden_host);
int mnf_rnvp_image_index(int dim, int n_hidden, const int* hidden_host, int32_t* idx_host);
/* Split image (see mnf_affine_half_split_layout): with image AND split_image given, both GEMMs run on
 * the f16 matrix pipe in split fp32 arithmetic; 128-row groups whose operands leave the f16 range are
 * recomputed with fp32 MFMAs inside the same launch. */
int mnf_rnvp_split_layout(int dim, int n_hidden, const int* hidden_host, int64_t* n_split_words,
                          int64_t* n_plain_words);
int mnf_rnvp_split_index(int dim, int n_hidden, const int* hidden_host, int32_t* idx_host);

/* ------------------------------------------------------- data-independent layers */
/* forward: y = x*exp(s)+t ; inverse: y = (x-t)*exp(-s).  s, t: (dim,) device vectors.
 * The (1,)-shaped log_det = +-sum(s) is a parameter-only scalar; ld_scalar (device, may be
 * NULL) receives it, and when log_det != NULL it is added to (accumulate) or broadcast
 * into (otherwise) every row. */
int mnf_affine_const(const float* x, float* y, const float* s, const float* t,
                     float* log_det, int accumulate, float* ld_scalar,
                     int64_t rows, int dim, int inverse, void* stream);
/* y = x @ W, W (dim, dim) row-major on the device (Glow's assembled matrix or its inverse). */
int mnf_linear_rows(const float* x, const float* W, float* y, int64_t rows, int dim, void* stream);

/* The same product with W pre-arranged as an MFMA operand image (dim in {16,32,64,128}):
 * image[i] = W_flat[idx[i]] with idx from mnf_linear_rows_image_index (dim*dim entries). */
int64_t mnf_linear_rows_image_floats(int dim);
int mnf_linear_rows_image_index(int dim, int32_t* idx_host);
int mnf_linear_rows_img(const float* x, const float* image, float* y, int64_t rows, int dim, void* stream);

/* --------------------------------------------------------- base log-prob epilogue */
/* log_prob[r] = (log_det ? log_det[r] : 0) - |z_r|^2/2 - dim/2*log(2 pi)   (standard normal base)
 * sum_out (device double, may be NULL) += sum_r log_prob[r]; the caller zeroes it first. */
int mnf_gauss_logprob(const float* z, const float* log_det, float* log_prob, double* sum_out,
                      int64_t rows, int dim, void* stream);

/* The same from precomputed z_sqnorm[r] = |z_r|^2 (see mnf_affine_half_sq). */
int mnf_gauss_logprob_sq(const float* z_sqnorm, const float* log_det, float* log_prob, double* sum_out,
                         int64_t rows, int dim, void* stream);

/* z0 = q0_mean + exp(q0_log_var)^(1/2) * eps   (rows, dim); mean, log_var: (dim,). */
int mnf_sample_z0(const float* q0_mean, const float* q0_log_var, const float* eps, float* z0,
                  int64_t rows, int dim, void* stream);
/* Gradients of that prologue: grad_mean[j] += sum_r grad_z0[r][j];  grad_log_var[j] += sum_r grad_z0[r][j] eps[r][j] *
 * 0.5 sqrt(exp(q0_log_var[j])).  Both outputs (dim floats each) are ADDED to. */
int mnf_sample_z0_bwd(const float* grad_z0, const float* eps, const float* q0_log_var, float* grad_mean,
                      float* grad_log_var, int64_t rows, int dim, void* stream);
/* The same two with the noise generated in place, eps[r][j] = a counter-based N(0, 1) of (seed, r, j) -- exactly what
 * mnf_sample_z0_noise(seed, eps, rows, dim) writes: `epsilon = torch.randn_like(...)` of mnf_linear.py:59-61 /
 * mnf_conv.py:91-93 without a (rows, dim) noise tensor being drawn, stored and read back by the gradient launch. */
int mnf_sample_z0_seeded(const float* q0_mean, const float* q0_log_var, uint64_t seed, float* z0, int64_t rows, int dim,
                         void* stream);
int mnf_sample_z0_seeded_bwd(const float* grad_z0, uint64_t seed, const float* q0_log_var, float* grad_mean,
                             float* grad_log_var, int64_t rows, int dim, void* stream);
int mnf_sample_z0_noise(uint64_t seed, float* eps, int64_t rows, int dim, void* stream);
/* mnf_sample_z0_bwd (eps != NULL) / mnf_sample_z0_seeded_bwd (eps == NULL) with the sums over the rows added in a fixed
 * order: every row block leaves its sums in `workspace` (mnf_sample_z0_bwd_workspace(rows, dim) floats) and a second
 * launch adds the blocks up in order -- no float atomics, the result repeats bit for bit (MNF_DETERMINISTIC=1). */
int64_t mnf_sample_z0_bwd_workspace(int64_t rows, int dim);
int mnf_sample_z0_bwd_det(const float* grad_z0, const float* eps, uint64_t seed, const float* q0_log_var, float* grad_mean,
                          float* grad_log_var, int64_t rows, int dim, float* workspace, int64_t workspace_floats,
                          void* stream);

/* ------------------------------------------------------------------ training: one optimizer launch
 * torch.optim.Adam's update (no amsgrad; weight_decay as L2 on the gradient) over ONE flat buffer: param, grad and
 * the two moment buffers are n floats each; step counts from 1.  The reference trains its flows with Adam
 * (tests/test_flows.py:41-50, examples/half_moons.ipynb:170-200); with the model's parameters re-homed in one buffer
 * (torch_mnf_amd.train.FlatParameters) the optimizer is one launch instead of one list entry per tensor. */
int mnf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream);
/* The same update with the step counter on the device, for an optimizer step captured in a hipGraph (a replay runs no
 * host code): state_dev = {step, 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} (3 floats, zero before the first
 * step); a one-thread kernel advances it, the update kernel behind it reads the two factors. */
int mnf_adam_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, float* state_dev, void* stream);

/* ------------------------------------------------------------------ NSF_AR (reuse of the spline device function)
 * torch_mnf/flows/spline_flow.py:182-235.  Element i is moved by a spline whose 3K-1 parameters come from
 * MLP_i(first i elements) -- of the OUTPUT in forward (inverse = 0; sequential in i; the spline runs inverted, :213-215)
 * and of the INPUT in inverse (:231-233); element 0 uses `init_param`.  log_det as for mnf_nsf_cl.
 * flat: init_param (3K-1), then the state_dict tensors of layers[0 .. dim-2] (layers[i-1] = MLP(i, hidden..., 3K-1)). */
int mnf_nsf_ar(const float* x, float* y, float* log_det, int accumulate, const float* flat, int64_t rows, int dim, int K,
               float tail_bound, int inverse, int n_hidden, const int* hidden_host, void* stream);
int64_t mnf_nsf_ar_flat_floats(int dim, int K, int n_hidden, const int* hidden_host);
int mnf_nsf_ar_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                   const float* flat, int64_t rows, int dim, int K, float tail_bound, int inverse, int n_hidden,
                   const int* hidden_host, void* stream);

/* ------------------------------------------------------------------ Glow: the d x d parameter preparation
 * torch_mnf/flows/glow.py:20-37.  out = W = P (tril(L,-1) + I) (triu(U,1) + diag(S)) (inverse = 0) or its inverse
 * (inverse = 1: two triangular inverses by substitution -- the matrix is given in PLU form -- instead of the
 * reference's torch.inverse of the assembled W, :34); log_det (1) = +-sum log|S| (:29, :35).  P, L, U (dim, dim), S (dim),
 * row-major, all device memory.  _bwd: grad_out = the cotangent of `out` (or NULL), grad_log_det (1) or NULL;
 * grad_L (strictly lower part; zeros elsewhere), grad_S, grad_U (strictly upper part) are written (accumulate = 0) or
 * ADDED to (accumulate = 1).  out_fwd (inverse = 1; or NULL): the `out` mnf_glow_weight produced for the same parameters --
 * the gradient launch then starts from that W^-1 instead of recomputing it (two substitutions and two products less).
 * dim <= MNF_GLOW_WEIGHT_MAX_DIM, else MNF_ERR_UNSUPPORTED (one workgroup, matrices in LDS). */
#define MNF_GLOW_WEIGHT_MAX_DIM 64
int mnf_glow_weight(const float* P, const float* L, const float* S, const float* U, float* out, float* log_det, int dim,
                    int inverse, void* stream);
int mnf_glow_weight_bwd(const float* P, const float* L, const float* S, const float* U, const float* grad_out,
                        const float* grad_log_det, float* grad_L, float* grad_S, float* grad_U, int dim, int inverse,
                        int accumulate, const float* out_fwd, void* stream);

/* ------------------------------------------------------------------ MAF / IAF (generic path)
 * torch_mnf/flows/maf.py:21-72 over net = MADE(dim, hidden, 2 dim, natural_ordering=True) (torch_mnf/layers/made.py:11-94:
 * MaskedLinear layers `x @ (W.T * mask) + b`, ReLU between them).
 *   flat   net.{0,2,..}.weight (n_out_l, n_in_l) | .bias, state_dict order          mnf_maf_flat_floats(...) floats
 *   masks  the MaskedLinear `mask` buffers (n_in_l, n_out_l) as bytes, back to back   mnf_maf_mask_bytes(...) bytes
 *   sequential = 0   one pass  (MAF.inverse, IAF.forward; maf.py:54-62): s, t = net(x); y = x exp(s) + t, flipped along
 *                    the features when parity; log_det = sum(s)
 *   sequential = 1   dim passes (MAF.forward, IAF.inverse; maf.py:39-52): the input is flipped first when parity; from
 *                    zeros, y_i = (z_i - t_i) exp(-s_i) with s, t = net(elements decoded so far); log_det = -sum(s_i)
 * log_det as everywhere (accumulate: +=).  mnf_maf_bwd: grad_x is written, grad_flat (layout of flat) is ADDED to (or
 * NULL); grad_y / grad_ld may be NULL; y = the forward call's output (sequential = 1 only, may be NULL otherwise).
 * One thread per row, masked weights and activations in LDS: MNF_ERR_UNSUPPORTED when a net does not fit 144 KB. */
int64_t mnf_maf_flat_floats(int dim, int n_hidden, const int* hidden_host);
int64_t mnf_maf_mask_bytes(int dim, int n_hidden, const int* hidden_host);
int mnf_maf(const float* x, float* y, float* log_det, int accumulate, const float* flat, const uint8_t* masks,
            int64_t rows, int dim, int parity, int sequential, int n_hidden, const int* hidden_host, void* stream);
int mnf_maf_bwd(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                const float* flat, const uint8_t* masks, int64_t rows, int dim, int parity, int sequential, int n_hidden,
                const int* hidden_host, void* stream);

/* ------------------------------------------------ MNFLinear.forward behind the flow path
 * torch_mnf/layers/mnf_linear.py:46-56 with z (rows, n_in) = what sample_z's last flow wrote:
 *   out = (x * z) W_mean^T + b_mean + sqrt(x^2 exp(W_log_var)^T + exp(b_log_var)) * eps       (rows, n_out), n_out <= 64
 * one pass, x and z read once.  eps (rows, n_out): the layer's N(0,1) noise, or NULL to have it
 * generated in-kernel from `seed` (mnf_mnf_linear_noise writes exactly those numbers).
 * flat (device): W_mean (n_out, n_in) | exp(W_log_var) * var_scale (n_out, n_in) | b_mean (n_out) | exp(b_log_var)
 * (n_out); var_scale a power of two that brings exp(W_log_var) into the f16 normal range, var_unscale = 1 / var_scale.
 * split_image: mnf_pack_gather_split of `flat` through the table of mnf_mnf_linear_split_index.
 * workspace: (rows + 127) / 128 int32 (written by the call: 128-row groups recomputed in fp32 by its fix-up launch). */
int mnf_mnf_linear_split_layout(int n_in, int n_out, int64_t* n_split_words, int64_t* n_plain_words);
int mnf_mnf_linear_split_index(int n_in, int n_out, int32_t* idx_host);
int mnf_mnf_linear_fwd(const float* x, const float* z, const float* eps, uint64_t seed, float* out, const float* flat,
                       const void* split_image, float var_unscale, int32_t* workspace, int64_t rows, int n_in, int n_out,
                       void* stream);
int mnf_mnf_linear_noise(uint64_t seed, float* eps, int64_t rows, int n_out, void* stream);
/* The training form of mnf_mnf_linear_fwd: additionally writes sd = sqrt(var) (rows, n_out) when sd_out != NULL (the
 * backward pass needs d out / d var = eps / (2 sd)); `workspace` then also carries the per-128-row range flags
 * mnf_mnf_linear_bwd reads.  Everything else as mnf_mnf_linear_fwd. */
int mnf_mnf_linear_fwd_train(const float* x, const float* z, const float* eps, uint64_t seed, float* out, float* sd_out,
                             const float* flat, const void* split_image, float var_unscale, int32_t* workspace,
                             int64_t rows, int n_in, int n_out, void* stream);
/* Gradients of MNFLinear.forward (layers/mnf_linear.py:46-56 under loss.backward(): what tests/test_mnf_mnist.py:14-56
 * trains through) on the matrix cores, n_out <= 64.  Two launches + an fp32 fix-up: a row-parallel prologue turns
 * grad_out, eps and sd into the cotangents of mean and var as split MFMA operands (1 KB per row in `workspace`) and
 * sums the bias gradients; a dims-slab launch (a workgroup owns 32 input dims and a range of rows) writes grad_x and
 * grad_z and sums dW_mean, dW_log_var over the rows in registers; 128-row groups flagged by the forward pass (or whose
 * cotangents leave the split range) are redone in fp32.
 *   flat            W_mean | exp(W_log_var) / var_unscale | b_mean | exp(b_log_var)   (the forward call's `flat`)
 *   bwd_image       mnf_pack_gather_split of `flat` through the mnf_mnf_linear_bwd_index table (see _layout)
 *   sd, fwd_flags   sd_out and workspace of the mnf_mnf_linear_fwd_train call; eps / seed as passed there
 *   grad_scale_dev  device float, a power of two that brings grad_out near 1 (mnf_affine_half_grad_scale)
 *   workspace       >= mnf_mnf_linear_bwd_workspace_bytes(rows, n_in, n_out) bytes, 16-byte aligned
 * grad_x, grad_z (rows, n_in) are written; grad_flat (layout of `flat`: d W_mean | d W_log_var | d b_mean |
 * d b_log_var) is ADDED to, or NULL. */
int64_t mnf_mnf_linear_bwd_workspace_bytes(int64_t rows, int n_in, int n_out);
int mnf_mnf_linear_bwd_layout(int n_in, int n_out, int64_t* n_split_words, int64_t* n_plain_words);
int mnf_mnf_linear_bwd_index(int n_in, int n_out, int32_t* idx_host);
int mnf_mnf_linear_bwd(const float* x, const float* z, const float* grad_out, const float* sd, const float* eps,
                       uint64_t seed, float* grad_x, float* grad_z, float* grad_flat, const float* flat,
                       const void* bwd_image, float var_unscale, const int32_t* fwd_flags, const float* grad_scale_dev,
                       void* workspace, int64_t workspace_bytes, int64_t rows, int n_in, int n_out, void* stream);

/* ------------------------------------------------ the MNF layers' KL term behind their flows
 * MNFLinear.kl_div (torch_mnf/layers/mnf_linear.py:66-90) and MNFConv2d.kl_div (torch_mnf/layers/mnf_conv.py:100-133)
 * with every random draw and both flows' results handed in: one launch forward, one backward (composed from stock
 * elementwise ops a layer's term is ~70 + ~100 launches).  The weight tensor is seen as a (rows, cols) matrix:
 *   conv = 0 (MNFLinear):  rows = n_out, cols = n_in, n_bias = n_out; eps (rows, cols) = the draw for `weight`
 *                          (mnf_linear.py:70-71); eps_b NULL; act = tanh(weight r0_c)
 *   conv = 1 (MNFConv2d):  rows = n_in k k, cols = n_out = n_bias (the `.view(-1, len(r0_c))` of mnf_conv.py:117-118
 *                          over the flat (n_out, n_in, k, k) tensor); eps (rows) = epsilon_w, eps_b (1) = epsilon_b;
 *                          linear act (:119-123); b_mean may be NULL (the zero buffer of :45)
 * z, z_r (cols): sample_z's z and flow_r's last output; log_det_q, log_det_r (1); q0_log_var, r0_c, r0_b1, r0_b2 (cols).
 * All pointers are device memory.  out (1) = kl_div_W + kl_div_b + log_q - log_r.
 * saved: mnf_mnf_kl_saved_floats(rows) floats written by _fwd and read by _bwd (act, its mean, the bias deviation).
 * _bwd, every entry times grad_out[0]:
 *   grads (written), mnf_mnf_kl_grad_floats(cols) floats:  dz (cols: the term's direct dependence; what reaches z
 *     through flow_r comes back through dz_r) | dz_r (cols) | dlog_det_q | dlog_det_r
 *   param_grads, mnf_mnf_kl_param_grad_floats(...) floats, in the order both reference modules register their
 *     parameters (mnf_linear.py:24-33, mnf_conv.py:43-57):  dW_mean (rows cols) | dW_log_var (rows cols) | db_mean
 *     (n_bias; conv = 0 only: MNFConv2d's b_mean is not a parameter) | db_log_var (n_bias) | dq0_mean (cols: no direct
 *     dependence) | dq0_log_var | dr0_c | dr0_b1 | dr0_b2 (cols each).  accumulate = 0: written (dq0_mean = 0);
 *     accumulate = 1: ADDED to -- a training loop that keeps all gradients in one buffer hands in the layer's slice.
 * rows > 28,672: MNF_ERR_UNSUPPORTED. */
int64_t mnf_mnf_kl_saved_floats(int64_t rows);
int64_t mnf_mnf_kl_grad_floats(int cols);
int64_t mnf_mnf_kl_param_grad_floats(int conv, int64_t rows, int cols, int n_bias);
int mnf_mnf_kl_fwd(const float* W_mean, const float* W_log_var, const float* eps, const float* eps_b, const float* z,
                   const float* z_r, const float* log_det_q, const float* log_det_r, const float* b_mean,
                   const float* b_log_var, const float* q0_log_var, const float* r0_c, const float* r0_b1,
                   const float* r0_b2, int conv, int64_t rows, int cols, int n_bias, float* out, float* saved,
                   void* stream);
int mnf_mnf_kl_bwd(const float* W_mean, const float* W_log_var, const float* eps, const float* eps_b, const float* z,
                   const float* z_r, const float* b_mean, const float* b_log_var, const float* r0_c, const float* r0_b1,
                   const float* r0_b2, const float* saved, const float* grad_out, int conv, int64_t rows, int cols,
                   int n_bias, float* grads, float* param_grads, int accumulate, void* stream);

/* ------------------------------------------------ the elementwise parts of MNFConv2d.forward
 * torch_mnf/layers/mnf_conv.py:67-88 around its two convolutions (which stay the caller's: MIOpen), one launch each way:
 *   _operands   Wz = W_mean * z[o] (the mean convolution's weight, :72),  W_var = exp(W_log_var),  b_var = exp(b_log_var)
 *               (:69-70); W_* are (n_out, per_out = n_in k k) row-major, z / b_* (n_out)
 *   _operands_bwd  from the cotangents of Wz, W_var, b_var (each may be NULL): grad_W_mean, grad_W_log_var,
 *               grad_b_log_var (written, or ADDED to when accumulate = 1) and grad_z (written)
 *   _noise      out = mean + sqrt(var) * eps (:86-88);  _noise_bwd: grad_var = grad_out * eps / (2 sqrt(var)) (the
 *               cotangent of mean is grad_out itself) */
int mnf_mnf_conv_operands(const float* W_mean, const float* W_log_var, const float* b_log_var, const float* z, float* Wz,
                          float* W_var, float* b_var, int n_out, int per_out, void* stream);
int mnf_mnf_conv_operands_bwd(const float* W_mean, const float* W_log_var, const float* b_log_var, const float* z,
                              const float* grad_Wz, const float* grad_W_var, const float* grad_b_var, float* grad_W_mean,
                              float* grad_W_log_var, float* grad_b_log_var, float* grad_z, int n_out, int per_out,
                              int accumulate, void* stream);
int mnf_mnf_noise(const float* mean, const float* var, const float* eps, float* out, int64_t n, void* stream);
int mnf_mnf_noise_bwd(const float* var, const float* eps, const float* grad_out, float* grad_var, int64_t n, void* stream);

/* ------------------------------------------------------------------ gradients (autograd)
 * What torch.autograd.Function.backward needs so the modules train like the reference's
 * (tests/test_flows.py:14-31 trains through forward/inverse).  grad_flat has the `flat` layout and
 * is ADDED to (the caller zeroes it); grad_y / grad_ld may be NULL (treated as zero). */
int mnf_affine_half_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                        float* grad_flat, const float* flat, int64_t rows, int dim, int parity,
                        int inverse, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                        void* stream);
/* The same gradients at matrix-pipe rate (fp32 MFMAs; three hidden layers of any widths <= 32 at even d <= 64, of any
 * widths <= 24 at even d <= 256: halves narrower than the kernel's 16/32/64/128-column tile and layers narrower than its
 * 16/24/32 unit slots are padded with structural zeros -- d = 2 runs here; a NICE-style layer with only one of the two
 * nets runs with the other as structural zeros too).  index_dev: device copy of the mnf_affine_half_bwd_index table
 * (mnf_affine_half_bwd_index_ints() int32s, built once per shape; 0 = no such kernel).  Same contract as
 * mnf_affine_half_bwd: grad_x is written, grad_flat is ADDED to; MNF_ERR_UNSUPPORTED otherwise. */
int64_t mnf_affine_half_bwd_index_ints(int dim, int n_hidden, const int* hidden_host, int has_scale, int has_shift);
int mnf_affine_half_bwd_index(int dim, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                              int32_t* idx_host);
int mnf_affine_half_bwd_mfma(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                             float* grad_flat, const float* flat, const int32_t* index_dev, int64_t rows, int dim,
                             int parity, int inverse, int n_hidden, const int* hidden_host, void* stream);
/* mnf_affine_half_bwd_mfma with the parameter sums added up in a fixed order (MNF_DETERMINISTIC=1): every workgroup
 * leaves its sums in `workspace` (mnf_affine_half_bwd_mfma_workspace(rows, ...) floats; 0 = no such kernel) and a second
 * launch adds the blocks in order -- no float atomics, results repeat bit for bit. */
int64_t mnf_affine_half_bwd_mfma_workspace(int64_t rows, int dim, int n_hidden, const int* hidden_host);
int mnf_affine_half_bwd_mfma_det(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                                 float* grad_flat, const float* flat, const int32_t* index_dev, int64_t rows, int dim,
                                 int parity, int inverse, int n_hidden, const int* hidden_host, float* workspace,
                                 int64_t workspace_floats, void* stream);
/* mnf_affine_half_bwd_mfma on the listed 16-row tiles only: tile_list_dev = [count, tile, tile, ...] on the device
 * (at most list_capacity tiles are read; count < 0: every tile); NULL = every tile.  The fix-up pass of
 * mnf_affine_half_bwd_split. */
int mnf_affine_half_bwd_mfma_tiles(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                                   float* grad_flat, const float* flat, const int32_t* index_dev, int64_t rows, int dim,
                                   int parity, int inverse, int n_hidden, const int* hidden_host,
                                   const int32_t* tile_list_dev, int list_capacity, void* stream);
/* The same gradients on the f16 matrix pipe in split form (mnf_split.h; same shapes as mnf_affine_half_bwd_mfma):
 * the training step's kernel (reference: every test in tests/test_flows.py:14-99 trains through these layers).
 *   bwd_image       split image of the forward AND the transposed operands: mnf_pack_gather_split with the
 *                   mnf_affine_half_bwd_split_index table (2 * n_split_words + n_plain_words int32s, see _layout);
 *                   to be repacked after a weight update
 *   index_dev       the mnf_affine_half_bwd_index table (flush order of the weight-gradient tiles)
 *   grad_scale_dev  device float, a power of two that brings the incoming gradients near 1 (they are ~1/rows for a
 *                   mean loss, below f16's normal range): mnf_affine_half_grad_scale writes one from a sample of up
 *                   to 512 rows spread evenly over the batch (row s * (rows / 512))
 *   cold_list       device int32 [1 + cold_capacity], cold_list[0] zeroed by the caller: 16-row tiles with an
 *                   operand outside the split range are appended and NOT accumulated (cold_list[0] = -1: all of
 *                   them -- weights beyond the split range, nothing was computed); the caller then runs
 *                   mnf_affine_half_bwd_mfma_tiles on that list (same stream).  cold_capacity >= ceil(rows / 16)
 *                   never overflows.
 *   workspace       optional device floats (mnf_affine_half_bwd_split_workspace() of them for the current device):
 *                   with it the parameter gradients are flushed in two stages (every workgroup stores its sums, a
 *                   second small launch adds them into grad_flat) instead of one atomic per parameter per
 *                   workgroup -- ~40 us less per launch whatever the row count; NULL / too small: atomics
 * grad_x is written, grad_flat ADDED to, as for the other gradient entry points. */
int mnf_affine_half_bwd_split_layout(int dim, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                                     int64_t* n_split_words, int64_t* n_plain_words);
int mnf_affine_half_bwd_split_index(int dim, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                                    int32_t* idx_host);
int mnf_affine_half_grad_scale(const float* grad_y, const float* grad_ld, int64_t rows, int dim, float* scale_out_dev,
                               void* stream);
int64_t mnf_affine_half_bwd_split_workspace(int64_t rows, int dim, int n_hidden, const int* hidden_host);
int mnf_affine_half_bwd_split(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                              float* grad_flat, const void* bwd_image, const int32_t* index_dev, int64_t rows, int dim,
                              int parity, int inverse, int n_hidden, const int* hidden_host,
                              const float* grad_scale_dev, int32_t* cold_list, int cold_capacity, float* workspace,
                              int64_t workspace_floats, void* stream);
/* mnf_affine_half_bwd_split for the LAST layer of a density pass under a standard-normal base, differentiating
 * log p = log_det + log N(y; 0, I) directly (the `loss.backward()` of `-model.log_prob(x).mean()`, the reference's
 * training loops: tests/test_flows.py:14-31 with flows/core.py:46-49): the cotangents are formed in the kernel from
 *   lp_grad      device floats (rows): d loss / d log p per row -- grad_ld = lp_grad, grad_y = -y lp_grad with the
 *                layer's output y recomputed from x (no `-z g` elementwise pass, no grad_y read);
 *   gy_scratch   device floats (rows x dim), 16-byte aligned, uninitialised: the rows of grad_y of every tile put on
 *                cold_list are stored there, for the mnf_affine_half_bwd_mfma_tiles fix-up (grad_y = gy_scratch,
 *                grad_ld = lp_grad).
 * mnf_affine_half_grad_scale(NULL, lp_grad, ...) gives a suitable grad_scale_dev.  Everything else as above. */
int mnf_affine_half_bwd_split_lp(const float* x, const float* lp_grad, float* gy_scratch, float* grad_x,
                                 float* grad_flat, const void* bwd_image, const int32_t* index_dev, int64_t rows, int dim,
                                 int parity, int inverse, int n_hidden, const int* hidden_host,
                                 const float* grad_scale_dev, int32_t* cold_list, int cold_capacity, float* workspace,
                                 int64_t workspace_floats, void* stream);
/* The same gradients for ANY conditioner shape on the f16 matrix pipe (mnf_ahf_bwd_rt.hip: run-time layer count and
 * widths, weights read from `flat`; no operand image, no index table): 1 .. 4 hidden layers of widths 4 .. 64, any even dim.
 * What the per-shape kernels above have no instantiation for runs here instead of on mnf_affine_half_bwd's VALU kernel.
 *   y               the layer's OUTPUT for the same x, direction and parameters (needed in the inverse direction with a
 *                   scale net: g_s = -grad_y y - grad_ld; may be NULL otherwise)
 *   grad_scale_dev  device float, a power of two that brings the cotangents near 1 (mnf_affine_half_grad_scale)
 * grad_x is written, grad_flat (layout of `flat`) is ADDED to with float atomics, one flush per block of 16 .. 128 rows (or
 * NULL: grad_x only).  MNF_ERR_UNSUPPORTED: shape outside these limits, or MNF_DETERMINISTIC is set (atomic sums). */
int mnf_affine_half_bwd_rt(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x,
                           float* grad_flat, const float* flat, const float* grad_scale_dev, int64_t rows, int dim,
                           int parity, int inverse, int n_hidden, const int* hidden_host, int has_scale, int has_shift,
                           void* stream);
int mnf_nsf_cl_bwd(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                   float* grad_flat, const float* flat, int64_t rows, int dim, int K, float tail_bound,
                   int inverse, int n_hidden, const int* hidden_host, void* stream);
/* The same gradients with the whole conditioner on the f16 matrix pipe in split arithmetic, a wave per 16-row tile
 * (mnf_nsf_bwd_tile.hip; NSF_CL under loss.backward(): torch_mnf/flows/spline_flow.py:249-285, tests/test_flows.py:89-99):
 * dim a multiple of 8 up to 64, three hidden layers of at most 16 units, K = 5 or 8 (K = 10 up to dim 32) -- the shapes
 * mnf_nsf_cl_bwd_tile_supported() answers 1 for.
 *   y            the layer's OUTPUT for the same x, direction and parameters (mnf_nsf_cl's): the second net's conditioner
 *                input is a column block of it, so the first half-step is not recomputed to get it
 *   bwd_image    mnf_pack_gather_split of `flat` through the table of mnf_nsf_cl_bwd_tile_index (n_split_words,
 *                n_plain_words from _layout; 2 * n_split_words + n_plain_words entries)
 *   flush_dev    the second table of _index on the device: n_params int32
 *   grad_scale_dev  device float, a power of two that brings the cotangents near 1 (mnf_affine_half_grad_scale)
 *   cold         device int32 [2 + 2 * cold_capacity], ZEROED by the caller, cold_capacity >= ceil(rows / 16): [0] the
 *                number of 16-row tiles whose operands left the split range (-1: the weights did), [1] != 0 when a gradient
 *                operand left f16's range, [2 ..] the tiles, [2 + cold_capacity ..] a flag per tile.  mnf_nsf_cl_bwd_tile_fixup (same stream, same arguments,
 *                `flat`) MUST follow: it recomputes exactly those rows -- every row in the last two cases -- on the
 *                generic fp32 kernel.
 *   workspace    mnf_nsf_cl_bwd_tile_workspace(...) floats: every workgroup's sums as one block; a second kernel adds
 *                the blocks to grad_flat in a fixed order (no float atomics: results repeat bit for bit). */
int mnf_nsf_cl_bwd_tile_supported(int dim, int K, int n_hidden, const int* hidden_host);
int mnf_nsf_cl_bwd_tile_layout(int dim, int K, int n_hidden, const int* hidden_host, int64_t* n_split_words,
                               int64_t* n_plain_words, int64_t* n_params);
int mnf_nsf_cl_bwd_tile_index(int dim, int K, int n_hidden, const int* hidden_host, int32_t* idx_host,
                              int32_t* flush_host);
int64_t mnf_nsf_cl_bwd_tile_workspace(int64_t rows, int dim, int K, int n_hidden, const int* hidden_host);
int mnf_nsf_cl_bwd_tile(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x, float* grad_flat,
                        const void* bwd_image, const int32_t* flush_dev, int64_t rows, int dim, int K, float tail_bound,
                        int inverse, int n_hidden, const int* hidden_host, const float* grad_scale_dev, int32_t* cold,
                        int cold_capacity, float* workspace, int64_t workspace_floats, void* stream);
int mnf_nsf_cl_bwd_tile_fixup(const float* x, const float* grad_y, const float* grad_ld, float* grad_x,
                              float* grad_flat, const float* flat, int64_t rows, int dim, int K, float tail_bound,
                              int inverse, int n_hidden, const int* hidden_host, const int32_t* cold,
                              int cold_capacity, void* stream);
/* mnf_nsf_cl_bwd for ANY dim, K = 2 .. 16 and 1 .. 4 hidden layers of widths 4 .. 64 on the f16 matrix pipe
 * (mnf_nsf_bwd_rt.hip: run-time shapes, weights read from `flat`; no operand image, no workspace).  y = the layer's OUTPUT
 * for the same x, direction and parameters; grad_scale_dev as for mnf_nsf_cl_bwd_tile.  grad_x is written, grad_flat ADDED
 * to with float atomics, one flush per block of 16 .. 128 rows and slot (or NULL).  MNF_ERR_UNSUPPORTED: shape outside these
 * limits, or MNF_DETERMINISTIC is set (atomic sums). */
int mnf_nsf_cl_bwd_rt(const float* x, const float* y, const float* grad_y, const float* grad_ld, float* grad_x,
                      float* grad_flat, const float* flat, const float* grad_scale_dev, int64_t rows, int dim, int K,
                      float tail_bound, int inverse, int n_hidden, const int* hidden_host, void* stream);
/* mask == NULL: the mask of the seeded forward call is regenerated from `seed`. */
int mnf_rnvp_bwd(const float* z, const float* mask, uint64_t seed, const float* grad_x,
                 const float* grad_ld, float* grad_z, float* grad_flat, const float* flat,
                 int64_t rows, int dim, int n_hidden, const int* hidden_host, void* stream);
/* The same gradients (flows/rnvp.py:25-39 under loss.backward(); what layers/mnf_linear.py:58-64,84 and
 * tests/test_mnf_mnist.py:14-56 train through) on the matrix cores in split arithmetic, for the shapes the split
 * forward kernels cover (one hidden layer of width <= 64, padded dim >= 64).  Two launches: a row-parallel one that
 * recomputes y, forms g_y and hands both over as MFMA operands (1 KB per row in `workspace`), and one in which a
 * workgroup owns 32 dims and a range of rows (grad_z; dWt, dWs, dWn, dbt, dbs as sums over rows in registers); row
 * groups outside the split range are redone by mnf_rnvp_bwd's kernel on those groups only.
 *   split_image     the FORWARD split image (mnf_rnvp_split_layout / _index + mnf_pack_gather_split)
 *   bwd_image       the backward-only image: mnf_pack_gather_split with the mnf_rnvp_bwd_mfma_index table
 *                   (2 * n_split_words + n_plain_words int32s, see _layout); repacked after a weight update
 *   grad_scale_dev  device float, a power of two that brings grad_x / grad_ld near 1 (mnf_affine_half_grad_scale)
 *   workspace       caller-owned device scratch of at least mnf_rnvp_bwd_mfma_workspace_bytes(rows, ...) bytes,
 *                   16-byte aligned (no need to clear it)
 * grad_x / grad_ld may be NULL (no cotangent for that output), grad_flat may be NULL (grad_z only).  grad_z is
 * written, grad_flat ADDED to.  MNF_ERR_UNSUPPORTED: no such kernel for the shape (use mnf_rnvp_bwd). */
int64_t mnf_rnvp_bwd_mfma_workspace_bytes(int64_t rows, int dim, int n_hidden, const int* hidden_host);
int mnf_rnvp_bwd_mfma_layout(int dim, int n_hidden, const int* hidden_host, int64_t* n_split_words,
                             int64_t* n_plain_words);
int mnf_rnvp_bwd_mfma_index(int dim, int n_hidden, const int* hidden_host, int32_t* idx_host);
int mnf_rnvp_bwd_mfma(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld,
                      float* grad_z, float* grad_flat, const float* flat, const void* split_image,
                      const void* bwd_image, const float* grad_scale_dev, void* workspace, int64_t workspace_bytes,
                      int64_t rows, int dim, int n_hidden, const int* hidden_host, void* stream);
/* The same call with its launches selectable (measurements: HIP events around one of them): phases bit 0 the
 * row-parallel launch A, bit 1 launch B-ts (grad_z, dWt, dWs), bit 2 launch B-n (dWn), bit 3 the fp32 fix-up; the
 * phases of one backward pass must run in that order on one stream.  15 = mnf_rnvp_bwd_mfma.
 * y: NULL, or the y = net(mask * z) the forward call kept (mnf_rnvp_seeded_train; rows x mnf_rnvp_y_floats_per_row
 * floats): launch A then skips its own sweep over z that recomputes it (a third of the launch: the Adam step of
 * MNFLinear(800, 50) at 256,000 rows 10.7 -> 10.1 ms).  Rows of NaN (row groups the forward pass recomputed in fp32)
 * send their group to this call's fp32 fix-up. */
int mnf_rnvp_bwd_mfma_phases(const float* z, const float* mask, uint64_t seed, const float* grad_x,
                             const float* grad_ld, float* grad_z, float* grad_flat, const float* flat,
                             const void* split_image, const void* bwd_image, const float* grad_scale_dev,
                             void* workspace, int64_t workspace_bytes, int64_t rows, int dim, int n_hidden,
                             const int* hidden_host, int phases, const float* y, void* stream);
/* mnf_rnvp_bwd for ANY conditioner shape on the f16 matrix pipe (mnf_rnvp_bwd_rt.hip: run-time layer count and widths,
 * weights read from `flat`; no operand image, no workspace): net = MLP(dim, h_1 .. h_n) with 1 .. 4 layers of widths
 * 4 .. 128, any dim.  grad_scale_dev: device float, a power of two that brings the cotangents near 1
 * (mnf_affine_half_grad_scale).  grad_z is written, grad_flat ADDED to with float atomics, one flush per block of 16 .. 128
 * rows (or NULL).  MNF_ERR_UNSUPPORTED: shape outside these limits, or MNF_DETERMINISTIC is set (atomic sums). */
int mnf_rnvp_bwd_rt(const float* z, const float* mask, uint64_t seed, const float* grad_x, const float* grad_ld, float* grad_z,
                    float* grad_flat, const float* flat, const float* grad_scale_dev, int64_t rows, int dim, int n_hidden,
                    const int* hidden_host, void* stream);
/* AffineConstantFlow: grad_x = grad_y * exp(+-s); grad_s, grad_t (dim,) are ADDED to. */
int mnf_affine_const_bwd(const float* x, const float* y, const float* grad_y, const float* s,
                         float* grad_x, float* grad_s, float* grad_t, int64_t rows, int dim,
                         int inverse, void* stream);
/* Glow.inverse followed by ActNormFlow.inverse (torch_mnf/flows/glow.py:33-37, affine_constant_flow.py:22-26: the pair
 * every [ActNormFlow, Glow, NSF_CL] block applies on the way x -> z) as one launch each way, dim = 16, 32 or 64 (else
 * MNF_ERR_UNSUPPORTED): z = (u @ M - t) e^-s with M = W^-1 (dim, dim) row-major and s, t (dim,) as the modules hold
 * them (no packed image: the kernels arrange M themselves).  The gradient launch writes grad_u and ADDS to grad_m
 * (dim, dim) = u^T (grad_z e^-s), grad_s, grad_t (dim,; either may be NULL); z is recomputed, not read.
 * ld_out (1, or NULL): the pair's log|det J| = ld_glow[0] (Glow's -sum log|S|, glow.py:35; NULL: 0) - sum s
 * (affine_constant_flow.py:25); grad_ld (1, or NULL): its cotangent, which enters grad_s as -grad_ld per column. */
int mnf_glow_actnorm_inv(const float* u, const float* M, const float* s, const float* t, float* z, const float* ld_glow,
                         float* ld_out, int64_t rows, int dim, void* stream);
int mnf_glow_actnorm_inv_bwd(const float* u, const float* grad_z, const float* M, const float* s, const float* t,
                             float* grad_u, float* grad_m, float* grad_s, float* grad_t, const float* grad_ld,
                             int64_t rows, int dim, void* stream);
/* The same pair closing a density pass under a standard-normal base (core.py:46-49: log p = log_det + base.log_prob(z),
 * distributions.py's Normal(0, 1)): log_prob[row] = log_det_rows[row] + (ld_glow[0] - sum s) - |z|^2 / 2 - dim log(2 pi) / 2
 * with z = (u @ M - t) e^-s never written.  The gradient launch takes grad_log_prob (rows,), forms grad_z = -z grad_log_prob
 * from the recomputed z, writes grad_u and ADDS to grad_m, grad_s, grad_t (as above) and to grad_ld_glow (1, or NULL) =
 * sum_r grad_log_prob[r]; the cotangent of log_det_rows is grad_log_prob itself. */
int mnf_glow_actnorm_inv_logprob(const float* u, const float* M, const float* s, const float* t, const float* ld_glow,
                                 const float* log_det_rows, float* log_prob, int64_t rows, int dim, void* stream);
int mnf_glow_actnorm_inv_logprob_bwd(const float* u, const float* grad_log_prob, const float* M, const float* s,
                                     const float* t, float* grad_u, float* grad_m, float* grad_s, float* grad_t,
                                     float* grad_ld_glow, int64_t rows, int dim, void* stream);
/* The two gradient launches above with their sums reduced in a FIXED order (MNF_DETERMINISTIC=1 in the Python layer; the
 * reference's training loop repeats bit for bit under its torch.manual_seed(0), tests/test_flows.py:11): every workgroup
 * stores its sums as one block of `workspace` (mnf_glow_actnorm_inv_bwd_workspace(rows, dim) floats) and a second kernel
 * adds the blocks up in block order -- no float atomics.  workspace == NULL: the atomic flush. */
int64_t mnf_glow_actnorm_inv_bwd_workspace(int64_t rows, int dim);
int mnf_glow_actnorm_inv_bwd_det(const float* u, const float* grad_z, const float* M, const float* s, const float* t,
                                 float* grad_u, float* grad_m, float* grad_s, float* grad_t, const float* grad_ld,
                                 int64_t rows, int dim, float* workspace, int64_t workspace_floats, void* stream);
int mnf_glow_actnorm_inv_logprob_bwd_det(const float* u, const float* grad_log_prob, const float* M, const float* s,
                                         const float* t, float* grad_u, float* grad_m, float* grad_s, float* grad_t,
                                         float* grad_ld_glow, int64_t rows, int dim, float* workspace,
                                         int64_t workspace_floats, void* stream);
/* Glow: grad_W (dim, dim) += x^T grad_y   (grad_x is mnf_linear_rows with W^T). */
int mnf_linear_rows_bwd_weight(const float* x, const float* grad_y, float* grad_W, int64_t rows, int dim,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNF_HIP_H */
