// Candidate-bitmap wire formats: the portable RoaringBitmap serialisation the
// reference stores and passes around (src/heed_codec/roaring_bitmap/
// roaring_bitmap_codec.rs:29-33) and its two milli-specific variants
// (cbo_roaring_bitmap_codec.rs, bo_roaring_bitmap_codec.rs). Host code only.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "vs_kernels.h"

namespace vs {

// Parses the header of a serialised RoaringBitmap and bounds-checks every
// container against `len`. Returns false on malformed input.
bool roaring_parse(const uint8_t* bytes, size_t len, std::vector<RoaringContainerDesc>* out);

// Sets the bits of every container into words[n_words] (docid d -> bit d & 63
// of word d >> 6); values beyond the mask are dropped. `words` must be zeroed.
// Returns the number of bits set.
uint64_t roaring_expand_host(const uint8_t* bytes, const std::vector<RoaringContainerDesc>& desc,
                             uint64_t* words, uint64_t n_words);

// Serialises the mask in the run-free form (cookie 12346): array containers up
// to 4096 values, bitmap containers above.
void roaring_serialize(const uint64_t* words, uint64_t n_words, std::vector<uint8_t>* out);

}  // namespace vs
