// Portable RoaringBitmap serialisation <-> dense docid mask, and the C-ABI
// entry points for the reference's three bitmap codecs.
//
// Layout of the portable format (all fields little-endian):
//   cookie   u32 = 12346, then u32 n_containers                 (no run containers)
//         or u32 = 12347 | (n_containers - 1) << 16, then a ceil(n/8)-byte
//            "container i is a run container" bitset
//   n x (u16 key, u16 cardinality - 1)
//   n x u32 byte offset of each container      (always with 12346; with 12347
//                                               only when n_containers >= 4)
//   containers: run    -> u16 n_runs, n_runs x (u16 start, u16 length - 1)
//               card > 4096 -> 1024 x u64 bitmap
//               otherwise   -> card x u16 sorted values
#include "vs_roaring.h"

#include <string.h>

#include "../../include/vs_abi.h"

namespace vs {

namespace {
constexpr uint32_t COOKIE_NO_RUN = 12346;
constexpr uint32_t COOKIE_RUN = 12347;
constexpr uint32_t NO_OFFSET_THRESHOLD = 4;

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline void wr16(std::vector<uint8_t>* o, uint32_t v) {
    o->push_back((uint8_t)v);
    o->push_back((uint8_t)(v >> 8));
}
inline void wr32(std::vector<uint8_t>* o, uint32_t v) {
    wr16(o, v & 0xFFFF);
    wr16(o, v >> 16);
}
}  // namespace

bool roaring_parse(const uint8_t* bytes, size_t len, std::vector<RoaringContainerDesc>* out) {
    out->clear();
    if (len < 8 && !(len >= 4 && (rd32(bytes) & 0xFFFF) == COOKIE_RUN)) return false;
    size_t pos = 0;
    const uint32_t cookie = rd32(bytes);
    pos += 4;
    uint32_t n = 0;
    const uint8_t* run_bits = nullptr;
    bool has_runs = false;
    if ((cookie & 0xFFFF) == COOKIE_RUN) {
        has_runs = true;
        n = (cookie >> 16) + 1;
        const size_t nb = (n + 7) / 8;
        if (pos + nb > len) return false;
        run_bits = bytes + pos;
        pos += nb;
    } else if (cookie == COOKIE_NO_RUN) {
        n = rd32(bytes + pos);
        pos += 4;
    } else {
        return false;
    }
    if (n > 65536) return false;
    if (pos + (size_t)n * 4 > len) return false;
    const uint8_t* descr = bytes + pos;
    pos += (size_t)n * 4;
    if (!has_runs || n >= NO_OFFSET_THRESHOLD) {
        if (pos + (size_t)n * 4 > len) return false;
        pos += (size_t)n * 4;  // the offsets are redundant: containers are laid out back to back
    }
    out->reserve(n);
    uint32_t prev_key = 0;
    for (uint32_t i = 0; i < n; ++i) {
        RoaringContainerDesc d;
        d.key = rd16(descr + 4 * i);
        if (i && d.key <= prev_key) return false;
        prev_key = d.key;
        const uint32_t card = rd16(descr + 4 * i + 2) + 1;
        const bool is_run = has_runs && ((run_bits[i >> 3] >> (i & 7)) & 1);
        if (is_run) {
            if (pos + 2 > len) return false;
            d.kind = 2;
            d.count = rd16(bytes + pos);
            pos += 2;
            d.offset = (uint32_t)pos;
            if (pos + (size_t)d.count * 4 > len) return false;
            for (uint32_t r = 0; r < d.count; ++r) {
                const uint32_t s = rd16(bytes + pos + 4 * r), l = rd16(bytes + pos + 4 * r + 2);
                if (s + l > 0xFFFF) return false;
            }
            pos += (size_t)d.count * 4;
        } else if (card > 4096) {
            d.kind = 1;
            d.count = card;
            d.offset = (uint32_t)pos;
            if (pos + 8192 > len) return false;
            pos += 8192;
        } else {
            d.kind = 0;
            d.count = card;
            d.offset = (uint32_t)pos;
            if (pos + (size_t)card * 2 > len) return false;
            pos += (size_t)card * 2;
        }
        out->push_back(d);
    }
    return true;
}

uint64_t roaring_expand_host(const uint8_t* bytes, const std::vector<RoaringContainerDesc>& desc,
                             uint64_t* words, uint64_t n_words) {
    uint64_t card = 0;
    for (const RoaringContainerDesc& d : desc) {
        const uint64_t word0 = (uint64_t)d.key << 10;
        const uint8_t* p = bytes + d.offset;
        auto set_bit = [&](uint32_t v) {
            const uint64_t wi = word0 + (v >> 6);
            if (wi < n_words && !((words[wi] >> (v & 63)) & 1)) {
                words[wi] |= 1ull << (v & 63);
                ++card;
            }
        };
        if (d.kind == 1) {
            for (uint32_t i = 0; i < 1024; ++i) {
                uint64_t v = 0;
                for (int b = 0; b < 8; ++b) v |= (uint64_t)p[i * 8 + b] << (8 * b);
                if (word0 + i < n_words) {
                    card += (uint64_t)__builtin_popcountll(v & ~words[word0 + i]);
                    words[word0 + i] |= v;
                }
            }
        } else if (d.kind == 0) {
            for (uint32_t i = 0; i < d.count; ++i) set_bit(rd16(p + 2 * i));
        } else {
            for (uint32_t r = 0; r < d.count; ++r) {
                const uint32_t s = rd16(p + 4 * r), e = s + rd16(p + 4 * r + 2);
                for (uint32_t v = s; v <= e; ++v) set_bit(v);
            }
        }
    }
    return card;
}

void roaring_serialize(const uint64_t* words, uint64_t n_words, std::vector<uint8_t>* out) {
    out->clear();
    struct C {
        uint32_t key, card;
    };
    std::vector<C> cs;
    const uint64_t n_chunks = (n_words + 1023) / 1024;
    for (uint64_t c = 0; c < n_chunks && c < 65536; ++c) {
        uint32_t card = 0;
        for (uint64_t i = c * 1024; i < (c + 1) * 1024 && i < n_words; ++i)
            card += (uint32_t)__builtin_popcountll(words[i]);
        if (card) cs.push_back({(uint32_t)c, card});
    }
    wr32(out, COOKIE_NO_RUN);
    wr32(out, (uint32_t)cs.size());
    for (const C& c : cs) {
        wr16(out, c.key);
        wr16(out, c.card - 1);
    }
    uint32_t off = 8 + 8 * (uint32_t)cs.size();
    for (const C& c : cs) {
        wr32(out, off);
        off += c.card > 4096 ? 8192 : 2 * c.card;
    }
    for (const C& c : cs) {
        const uint64_t w0 = (uint64_t)c.key * 1024;
        if (c.card > 4096) {
            for (uint64_t i = w0; i < w0 + 1024; ++i) {
                const uint64_t v = i < n_words ? words[i] : 0;
                for (int b = 0; b < 8; ++b) out->push_back((uint8_t)(v >> (8 * b)));
            }
        } else {
            for (uint64_t i = w0; i < w0 + 1024 && i < n_words; ++i) {
                uint64_t v = words[i];
                while (v) {
                    const int b = __builtin_ctzll(v);
                    wr16(out, (uint32_t)((i - w0) * 64 + b));
                    v &= v - 1;
                }
            }
        }
    }
}

namespace {
uint64_t raw_u32_list_to_mask(const uint8_t* bytes, size_t len, uint64_t* words, uint64_t n_words) {
    uint64_t card = 0;
    for (size_t i = 0; i + 4 <= len; i += 4) {
        uint32_t v;
        memcpy(&v, bytes + i, 4);  // native endian
        if ((uint64_t)(v >> 6) < n_words && !((words[v >> 6] >> (v & 63)) & 1)) {
            words[v >> 6] |= 1ull << (v & 63);
            ++card;
        }
    }
    return card;
}
int32_t copy_out(const std::vector<uint8_t>& v, uint8_t* out, size_t cap, size_t* len) {
    if (len) *len = v.size();
    if (!out) return VS_OK;
    if (cap < v.size()) return VS_E_CAPACITY;
    if (!v.empty()) memcpy(out, v.data(), v.size());
    return VS_OK;
}
constexpr size_t CBO_THRESHOLD = 7;  // cbo_roaring_bitmap_codec.rs:15
}  // namespace

}  // namespace vs

using namespace vs;

extern "C" int32_t vs_roaring_to_mask(const uint8_t* bytes, size_t len, uint64_t* words,
                                      uint64_t n_words, uint64_t* cardinality) {
    if (!bytes || (!words && n_words)) return VS_E_INVALID_ARG;
    std::vector<RoaringContainerDesc> desc;
    if (!roaring_parse(bytes, len, &desc)) return VS_E_FORMAT;
    if (n_words) memset(words, 0, n_words * 8);
    const uint64_t card = roaring_expand_host(bytes, desc, words, n_words);
    if (cardinality) *cardinality = card;
    return VS_OK;
}

extern "C" int32_t vs_mask_to_roaring(const uint64_t* words, uint64_t n_words, uint8_t* out,
                                      size_t cap, size_t* len) {
    if (!words && n_words) return VS_E_INVALID_ARG;
    std::vector<uint8_t> v;
    roaring_serialize(words, n_words, &v);
    return copy_out(v, out, cap, len);
}

extern "C" int32_t vs_cbo_roaring_to_mask(const uint8_t* bytes, size_t len, uint64_t* words,
                                          uint64_t n_words, uint64_t* cardinality) {
    if ((!bytes && len) || (!words && n_words)) return VS_E_INVALID_ARG;
    if (len <= CBO_THRESHOLD * 4) {
        if (n_words) memset(words, 0, n_words * 8);
        const uint64_t card = raw_u32_list_to_mask(bytes, len, words, n_words);
        if (cardinality) *cardinality = card;
        return VS_OK;
    }
    return vs_roaring_to_mask(bytes, len, words, n_words, cardinality);
}

extern "C" int32_t vs_mask_to_cbo_roaring(const uint64_t* words, uint64_t n_words, uint8_t* out,
                                          size_t cap, size_t* len) {
    if (!words && n_words) return VS_E_INVALID_ARG;
    uint64_t card = 0;
    for (uint64_t i = 0; i < n_words; ++i) card += (uint64_t)__builtin_popcountll(words[i]);
    if (card > CBO_THRESHOLD) return vs_mask_to_roaring(words, n_words, out, cap, len);
    std::vector<uint8_t> v;
    for (uint64_t i = 0; i < n_words; ++i) {
        uint64_t w = words[i];
        while (w) {
            const uint32_t d = (uint32_t)(i * 64 + __builtin_ctzll(w));
            uint8_t b[4];
            memcpy(b, &d, 4);  // native endian
            v.insert(v.end(), b, b + 4);
            w &= w - 1;
        }
    }
    return copy_out(v, out, cap, len);
}

extern "C" int32_t vs_bo_roaring_to_mask(const uint8_t* bytes, size_t len, uint64_t* words,
                                         uint64_t n_words, uint64_t* cardinality) {
    if ((!bytes && len) || (!words && n_words)) return VS_E_INVALID_ARG;
    if (len % 4) return VS_E_FORMAT;  // bytes_decode: chunk.try_into()? fails on a short chunk
    if (n_words) memset(words, 0, n_words * 8);
    const uint64_t card = raw_u32_list_to_mask(bytes, len, words, n_words);
    if (cardinality) *cardinality = card;
    return VS_OK;
}
