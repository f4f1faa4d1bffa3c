/*
 * az_net.h — C ABI of the fused PV-network inference path (gfx950 MFMA kernels).
 *
 * Replaces, for the self-play hot path, `Net.forward` of the reference (network.py:48-64,
 * ResidualBlock.forward network.py:99-104) as it is called by the inference server
 * (`handle_gpu`, examplegenerator.py:72-77): a batch of (C+1,H,W) boards in, softmax priors
 * [B,A] and tanh values [B] out — all device resident, so az_engine_advance() and
 * az_net_forward() ping-pong on one stream without touching the host.
 *
 * The weights are handed over PRE-PACKED by the host side (alphazero-openspiel_amd/fusednet.py
 * documents and tests the packing): eval-mode BatchNorm folded (bn2 into conv1's weights/bias, bn1
 * kept as a per-channel scale/shift prologue), 3x3 convs laid out as MFMA A-fragments of an implicit
 * GEMM with K = 64 groups x 8 channels (9 taps x 7 channel-groups; group 63 is zero padding), fp16
 * operands, fp32 accumulation, fp32 residual stream; block 1's 1x1 skip conv (network.py:96-97,102-103)
 * is applied in fp32 straight into the residual stream.
 *
 * Same conventions as az_engine.h: int status returns (0 ok / negative AZ_E_*), no exceptions,
 * az_net_last_error() for text, `stream` = hipStream_t as void*.
 */
#ifndef AZ_NET_H
#define AZ_NET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic of the matrix products (both accumulate in fp32; the residual stream, biases and BatchNorm affines are fp32):
 *   AZ_NET_PREC_F16    fp16 operands: one MFMA per product.  |dprior| <= 4e-3, |dvalue| <= 8e-3 against the reference's
 *                      fp32 Net.forward; search-level tolerance measured in tests/test_precision_search_gpu.py.
 *   AZ_NET_PREC_F16X3  fp32-grade: every operand is a pair of fp16 numbers (hi + lo / 2048), three MFMAs per product
 *                      (hi*hi + hi*lo + lo*hi), ~22 mantissa bits per product.  The reference's arithmetic is fp32
 *                      (network.py:48-64); gfx950's f32-input MFMA runs at 1/16 of the fp16 rate, this runs at 1/3. */
#define AZ_NET_PREC_F16 0
#define AZ_NET_PREC_F16X3 1

#define AZ_NET_CPAD 56      /* channels per LDS cell: 7 groups of 8 (n_filters <= 56) */
#define AZ_NET_KSTEPS 16    /* 32-deep MFMA k-steps per conv: 64 groups x 8 */
#define AZ_NET_XOUT_C 64    /* channel stride of the tower output handed to the FC kernel */

typedef struct az_net_desc {
    int32_t struct_size;
    int32_t rows, cols;     /* H, W of the board */
    int32_t in_planes;      /* C+1 = 4 */
    int32_t n_filters;      /* 50 in the reference (network.py:37) */
    int32_t n_blocks;       /* 5 in the reference (network.py:39-43) */
    int32_t num_actions;    /* A; fc1 has A+1 outputs (network.py:45) */
    int32_t device;
    int32_t precision;      /* AZ_NET_PREC_* */
    int32_t reserved;
    /* host pointers to the packed parameters (copied by az_net_create) */
    const uint16_t *conv_w; /* fp16 bits [2*n_blocks][16 ksteps][4 mtiles][64 lanes][8]: the interchange layout (group g = tap * 7 + channel
                             * octet); az_net_create re-groups it for the device (conv 0: 4 k-steps; <= 50 filters: 15 k-steps) */
    const float *conv_epi;  /* [2*n_blocks][3][64]: bias, next-prologue scale, next-prologue shift */
    const float *in_affine; /* [2][8]: block-1 bn1 scale / shift for the input planes */
    const float *skip_w;    /* [64][4]: block-1 conv3 (1x1) weights, out-channel major, zero padded */
    const uint16_t *fc_w;   /* fp16 bits [n_otiles][H*W*64/32 ksteps][64 lanes][8] */
    const float *fc_b;      /* [n_otiles*16] (bias of fc1, zero padded) */
    /* AZ_NET_PREC_F16X3 only: the "lo" halves, fp16 bits of (w - fp16(w)) * 2048, in the layouts of conv_w / fc_w */
    const uint16_t *conv_w_lo;
    const uint16_t *fc_w_lo;
} az_net_desc;

typedef struct az_net az_net;

int az_net_create(const az_net_desc *desc, az_net **out);
int az_net_destroy(az_net *n);
const char *az_net_last_error(const az_net *n);

/* obs: dev float32 [n_boards][in_planes][H][W] (what az_engine_advance wrote);
 * priors: dev float32 [n_boards][A] = softmax(fc1[:A]); values: dev float32 [n_boards] = tanh(fc1[A])
 * (network.py:61-64).  n_boards <= max_boards given at create time via az_net_reserve. */
int az_net_forward(az_net *n, const float *obs, float *priors, float *values, int32_t n_boards, void *stream);

/* (Re)allocate the intermediate tower-output buffer for up to max_boards boards. */
int az_net_reserve(az_net *n, int32_t max_boards);

/* Debug/parity: copy the tower output of the last forward (pre-fc residual stream, fp16 as float32,
 * [n_boards][H*W][64]) to a host buffer.  Synchronises the device. */
int az_net_read_tower(az_net *n, float *out, int32_t n_boards);

/* Measurement aid (bench.py's roofline.frac_issued): the number of v_mfma_f32_16x16x32_f16 instructions (16384 FLOP each,
 * padding included) the tower + head kernels issue per board for a launch of n_boards boards, as launched by az_net_forward
 * (the work partition depends on the batch size).  Host arithmetic only; touches no device. */
int az_net_issued_mfma_per_board(const az_net *n, int32_t n_boards, double *out);

/* Names of the kernels az_net_forward launches for a batch of n_boards boards of this net ("az_tower_x3b_kernel +
 * az_head_kernel<X3>", ...; the choice depends on the batch size, like the count above; n_boards < 1: the reserved maximum):
 * labels for bench.py's roofline object and for matching rocprofv3 rows.  Static storage. */
const char *az_net_kernel_label(const az_net *n, int32_t n_boards);

#ifdef __cplusplus
}
#endif
#endif /* AZ_NET_H */
