/* lookonce_weights.h — on-disk format of the packed weight blob (`python -m lookoncetohear_amd.checkpoint export`).
 *
 * Adjacent to the hot path: the reference keeps weights in a Lightning checkpoint
 * (`torch.load(run_dir/best.ckpt)['state_dict']`, /root/reference/src/ts_hear_test.py:18-26); a host that drives the
 * C ABI of lookonce_hip.h without Python reads this blob instead.  Every tensor is stored exactly as the kernels
 * consume it (MFMA fragment order, fp16 hi/lo images, LayerNorm affines folded), 256-byte aligned, so the payload is
 * uploaded with one hipMemcpy and addressed as `device_base + offset`.
 *
 *   bytes 0..7        magic "LHWPACK1"
 *   bytes 8..11       uint32 LE   ABI version the images were packed for (must equal lh_abi_version())
 *   bytes 12..15      uint32 LE   length L of the JSON index
 *   bytes 16..16+L    JSON        {"model": "separator" | "embedder", "params": {constructor keywords},
 *                                  "payload_bytes": N,
 *                                  "tensors": [{"name", "dtype", "shape", "offset", "nbytes"}, ...]}
 *   zero padding to the next multiple of 256, then N payload bytes
 *
 * Tensor names are the keys of weights.pack_all / embed_net.pack_embedder flattened with '.', e.g.
 * "blocks.0.intra_w16" (argument `w_pk` of lh_intra_block), "wfb_t" (argument `wfb_pk` of lh_stft_conv_in).
 */
#ifndef LOOKONCE_WEIGHTS_H
#define LOOKONCE_WEIGHTS_H

#include <stdint.h>

#define LH_WEIGHTS_MAGIC "LHWPACK1"
#define LH_WEIGHTS_ALIGN 256

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lh_weights_header {
    char magic[8];        /* LH_WEIGHTS_MAGIC, not NUL-terminated */
    uint32_t abi_version; /* little endian */
    uint32_t index_bytes; /* length of the JSON index that follows */
} lh_weights_header;

/* byte offset of the payload from the start of the file */
static inline uint64_t lh_weights_payload_offset(const lh_weights_header* h) {
    const uint64_t end = 16u + (uint64_t)h->index_bytes;
    return (end + LH_WEIGHTS_ALIGN - 1) / LH_WEIGHTS_ALIGN * LH_WEIGHTS_ALIGN;
}

#ifdef __cplusplus
}
#endif
#endif /* LOOKONCE_WEIGHTS_H */
