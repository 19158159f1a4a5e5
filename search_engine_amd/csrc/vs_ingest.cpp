// Filling stores from the reference's own vector database: raw (key, value) pairs of the
// `vector-arroy` LMDB database (src/index.rs:155,208) as a host reads them with a cursor.
//
// arroy 0.2.0 is not in the reference tree; its on-disk layout is restated here FROM MEMORY of the
// published crate (UNVERIFIED) and kept in two functions, arroy_parse_key and arroy_parse_leaf, so
// that a correction touches nothing else:
//   key   = index: u16 big-endian | mode: u8 (0 item, 1 tree, 2 metadata) | item: u32 big-endian
//   item  = tag: u8 (0 leaf; 1 descendants and 2 split-plane nodes live under tree keys)
//           | Angular header { norm: f32 } | vector: dim x f32, native (little) endian, unaligned
// The leaf's norm is not taken over: vs_store_seal recomputes ‖row‖ with the scan's own summation
// order, which is what keeps a search's distances independent of how a row arrived.
#include <string.h>

#include <map>
#include <vector>

#include "vs_internal.h"

using vsrt::fail;

namespace {

constexpr size_t ARROY_KEY_BYTES = 7;
enum : uint8_t { ARROY_MODE_ITEM = 0, ARROY_MODE_TREE = 1, ARROY_MODE_METADATA = 2 };
constexpr uint8_t ARROY_LEAF_TAG = 0;
constexpr size_t ARROY_LEAF_HEADER_BYTES = 1 + 4;  // tag + Angular's norm

bool arroy_parse_key(const uint8_t* key, size_t len, uint16_t* index, uint8_t* mode, uint32_t* item) {
    if (!key || len != ARROY_KEY_BYTES) return false;
    *index = (uint16_t)((uint16_t)key[0] << 8 | key[1]);
    *mode = key[2];
    *item = (uint32_t)key[3] << 24 | (uint32_t)key[4] << 16 | (uint32_t)key[5] << 8 | key[6];
    return true;  // (a mode this layout does not know is some other node: only item leaves are read)
}

// The vector of an item leaf: pointer into `value` (unaligned) and its dimension; false if the
// value is no leaf.
bool arroy_parse_leaf(const uint8_t* value, size_t len, const uint8_t** vector, uint32_t* dim) {
    if (!value || len < ARROY_LEAF_HEADER_BYTES || value[0] != ARROY_LEAF_TAG) return false;
    if ((len - ARROY_LEAF_HEADER_BYTES) % 4) return false;
    *vector = value + ARROY_LEAF_HEADER_BYTES;
    *dim = (uint32_t)((len - ARROY_LEAF_HEADER_BYTES) / 4);
    return true;
}

}  // namespace

extern "C" int32_t vs_arroy_parse_key(const uint8_t* key, size_t len, uint16_t* index, uint8_t* mode,
                                      uint32_t* item) {
    if (!index || !mode || !item) return fail(VS_E_INVALID_ARG, "null output");
    if (!arroy_parse_key(key, len, index, mode, item))
        return fail(VS_E_FORMAT, "not an arroy key (%zu bytes)", len);
    return VS_OK;
}

extern "C" int32_t vs_arroy_encode_item(uint16_t index, uint32_t item, const float* vector, uint32_t dim,
                                        uint8_t* key_out, uint8_t* value_out, size_t value_cap, size_t* value_len) {
    if (!vector || !key_out || !value_len) return fail(VS_E_INVALID_ARG, "null argument");
    const size_t need = ARROY_LEAF_HEADER_BYTES + (size_t)dim * 4;
    *value_len = need;
    key_out[0] = (uint8_t)(index >> 8);
    key_out[1] = (uint8_t)index;
    key_out[2] = ARROY_MODE_ITEM;
    key_out[3] = (uint8_t)(item >> 24);
    key_out[4] = (uint8_t)(item >> 16);
    key_out[5] = (uint8_t)(item >> 8);
    key_out[6] = (uint8_t)item;
    if (!value_out) return VS_OK;
    if (value_cap < need) return fail(VS_E_CAPACITY, "leaf of %zu bytes does not fit %zu", need, value_cap);
    value_out[0] = ARROY_LEAF_TAG;
    // the header's norm as add_item stores it: sqrt of the plain f32 dot product
    float acc = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) acc += vector[i] * vector[i];
    const float norm = __builtin_sqrtf(acc);
    memcpy(value_out + 1, &norm, 4);
    memcpy(value_out + ARROY_LEAF_HEADER_BYTES, vector, (size_t)dim * 4);
    return VS_OK;
}

extern "C" int32_t vs_ingest_arroy_items(vs_ctx* ctx, const uint8_t* const* keys, const size_t* key_lens,
                                         const uint8_t* const* values, const size_t* value_lens, uint64_t n,
                                         uint64_t* n_items, uint64_t* n_other_nodes, uint64_t* n_unrouted) {
    if (!ctx) return fail(VS_E_INVALID_ARG, "null context");
    if (n && (!keys || !key_lens || !values || !value_lens)) return fail(VS_E_INVALID_ARG, "null pair arrays");
    uint64_t items = 0, other = 0, unrouted = 0;
    if (n_items) *n_items = 0;
    if (n_other_nodes) *n_other_nodes = 0;
    if (n_unrouted) *n_unrouted = 0;
    struct Batch {
        vs_store* store;
        std::vector<uint32_t> docids;
        std::vector<float> rows;
    };
    std::map<uint16_t, Batch> batches;
    auto flush = [&](Batch& b) -> int32_t {
        if (b.docids.empty()) return VS_OK;
        const int32_t rc = vs_store_upload(b.store, b.docids.data(), b.rows.data(), b.docids.size());
        b.docids.clear();
        b.rows.clear();
        return rc;
    };
    auto store_of = [&](uint16_t index) -> Batch& {
        auto it = batches.find(index);
        if (it == batches.end()) {
            vs_store* store = nullptr;
            {
                std::lock_guard<std::mutex> g(ctx->mu);
                auto st = ctx->stores.find(index);
                if (st != ctx->stores.end()) store = st->second;
            }
            it = batches.emplace(index, Batch{store, {}, {}}).first;
        }
        return it->second;
    };
    // First every pair is checked, then rows are uploaded: a malformed pair or a leaf of the wrong
    // dimension fails the call before any store has been touched.
    for (uint64_t i = 0; i < n; ++i) {
        uint16_t index;
        uint8_t mode;
        uint32_t item;
        if (!arroy_parse_key(keys[i], key_lens[i], &index, &mode, &item))
            return fail(VS_E_FORMAT, "pair %llu: not an arroy key (%zu bytes)", (unsigned long long)i, key_lens[i]);
        if (mode != ARROY_MODE_ITEM) continue;
        Batch& b = store_of(index);
        if (!b.store) continue;
        const uint8_t* vec;
        uint32_t dim;
        if (!arroy_parse_leaf(values[i], value_lens[i], &vec, &dim))
            return fail(VS_E_FORMAT, "pair %llu: item %u of index %u is no leaf (%zu bytes)", (unsigned long long)i,
                        item, index, value_lens[i]);
        if (dim != b.store->dim) {
            vsrt::set_dim_mismatch(b.store->dim, dim);
            return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", b.store->dim, dim);
        }
    }
    for (uint64_t i = 0; i < n; ++i) {
        uint16_t index;
        uint8_t mode;
        uint32_t item;
        (void)arroy_parse_key(keys[i], key_lens[i], &index, &mode, &item);
        if (mode != ARROY_MODE_ITEM) {  // tree nodes, metadata, modes of other arroy versions: an exhaustive scan uses no forest
            ++other;
            continue;
        }
        Batch& b = store_of(index);
        if (!b.store) {  // an index the host made no store for
            ++unrouted;
            continue;
        }
        const uint8_t* vec;
        uint32_t dim;
        (void)arroy_parse_leaf(values[i], value_lens[i], &vec, &dim);
        b.docids.push_back(item);
        const size_t at = b.rows.size();
        b.rows.resize(at + dim);
        memcpy(b.rows.data() + at, vec, (size_t)dim * 4);
        ++items;
        if (b.rows.size() * 4 >= (16u << 20))
            if (int32_t rc = flush(b)) return rc;
    }
    for (auto& kv : batches)
        if (int32_t rc = flush(kv.second)) return rc;
    if (n_items) *n_items = items;
    if (n_other_nodes) *n_other_nodes = other;
    if (n_unrouted) *n_unrouted = unrouted;
    return VS_OK;
}

extern "C" int32_t vs_store_writer_len(vs_store* s, uint64_t* n_items) {
    if (!s || !n_items) return fail(VS_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> g(s->ctx->upload_mu);
    *n_items = s->n_items;
    return VS_OK;
}
