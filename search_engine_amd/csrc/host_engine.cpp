// C++ mirror of the reference's host logic on the vector-search path; see
// host_engine.h for the map onto the reference's files. CPU control flow only.
#include "host_engine.h"

#include <cstdlib>
#include <unordered_map>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstring>
#include <limits>

#include "../../include/vs_abi.h"

namespace search_engine {

// ---- DocSet -------------------------------------------------------------------------

namespace {
inline uint64_t popcount64(uint64_t x) { return (uint64_t)__builtin_popcountll(x); }
uint64_t count_bits(const std::vector<uint64_t>& w) {
    uint64_t n = 0;
    for (uint64_t x : w) n += popcount64(x);
    return n;
}
std::vector<uint64_t> range_words(uint64_t n) {
    std::vector<uint64_t> w((size_t)((n + 63) / 64), ~0ull);
    if (n & 63) w.back() = (1ull << (n & 63)) - 1;
    return w;
}
}  // namespace

DocSet DocSet::from_ids(const std::vector<DocumentId>& ids) {
    DocSet s;
    if (ids.size() <= SMALL_MAX) {
        s.ids_ = ids;
        std::sort(s.ids_.begin(), s.ids_.end());
        s.ids_.erase(std::unique(s.ids_.begin(), s.ids_.end()), s.ids_.end());
        return s;
    }
    DocumentId top = 0;
    for (DocumentId d : ids) top = std::max(top, d);
    auto block = std::make_shared<Block>(std::vector<uint64_t>((size_t)(top >> 6) + 1, 0));
    uint64_t n = 0;
    for (DocumentId d : ids) {
        uint64_t& w = block->w[d >> 6];
        const uint64_t bit = 1ull << (d & 63);
        n += !(w & bit);
        w |= bit;
    }
    if (n == (uint64_t)top + 1) return from_range(n);  // every docid up to the largest: no holes
    block->len = (int64_t)n;
    s.kind_ = Bitmap;
    s.dense_ = std::move(block);
    return s;
}
DocSet DocSet::from_words(std::vector<uint64_t> words) {
    DocSet s;
    s.kind_ = Bitmap;
    s.dense_ = std::make_shared<Block>(std::move(words));
    return s;
}
DocSet DocSet::from_range(uint64_t n) {
    DocSet s;
    if (n) {
        s.kind_ = Range;
        s.range_n_ = n;
    }
    return s;
}
const char* DocSet::representation() const {
    return kind_ == Array ? "array" : kind_ == Range ? "range" : "bitmap";
}
void DocSet::to_bitmap() {
    if (kind_ == Bitmap) return;
    std::vector<uint64_t> w;
    if (kind_ == Range) {
        w = range_words(range_n_);
    } else if (!ids_.empty()) {
        w.assign((size_t)(ids_.back() >> 6) + 1, 0);
        for (DocumentId d : ids_) w[d >> 6] |= 1ull << (d & 63);
    }
    const int64_t n = kind_ == Range ? (int64_t)range_n_ : (int64_t)ids_.size();
    dense_ = std::make_shared<Block>(std::move(w));
    dense_->len = n;
    kind_ = Bitmap;
    ids_.clear();
    range_n_ = 0;
    cache_.reset();
}
std::vector<uint64_t>& DocSet::own() {
    if (!dense_) {
        dense_ = std::make_shared<Block>();
        dense_->len = 0;
    } else if (dense_.use_count() > 1) {
        auto mine = std::make_shared<Block>(dense_->w);
        mine->len = dense_->len.load();
        dense_ = std::move(mine);
    } else {
        std::lock_guard<std::mutex> g(dense_->mu);
        dense_->attached.clear();  // whatever was derived from the words is stale from here on
    }
    return dense_->w;
}
const DocSet::Block& DocSet::word_block() const {
    static const Block none;
    if (kind_ == Bitmap) return dense_ ? *dense_ : none;
    if (!cache_) {
        DocSet copy = *this;
        copy.cache_.reset();
        copy.to_bitmap();
        cache_ = copy.dense_;
    }
    return *cache_;
}
const std::vector<uint64_t>& DocSet::words() const { return word_block().w; }
std::shared_ptr<void> DocSet::attachment(const void* owner) const {
    Block& b = const_cast<Block&>(word_block());
    std::lock_guard<std::mutex> g(b.mu);
    for (auto& kv : b.attached)
        if (kv.first == owner) return kv.second;
    return nullptr;
}
void DocSet::attach(const void* owner, std::shared_ptr<void> what) const {
    Block& b = const_cast<Block&>(word_block());
    std::lock_guard<std::mutex> g(b.mu);
    for (auto& kv : b.attached)
        if (kv.first == owner) {
            kv.second = std::move(what);
            return;
        }
    b.attached.emplace_back(owner, std::move(what));
}
bool DocSet::contains(DocumentId d) const {
    if (kind_ == Array) return std::binary_search(ids_.begin(), ids_.end(), d);
    if (kind_ == Range) return (uint64_t)d < range_n_;
    const size_t wi = d >> 6;
    return dense_ && wi < dense_->w.size() && ((dense_->w[wi] >> (d & 63)) & 1);
}
bool DocSet::insert(DocumentId d) {
    if (kind_ == Array) {
        auto it = std::lower_bound(ids_.begin(), ids_.end(), d);
        if (it != ids_.end() && *it == d) return false;
        if (ids_.size() < SMALL_MAX) {
            ids_.insert(it, d);
            cache_.reset();
            return true;
        }
        to_bitmap();
    } else if (kind_ == Range) {
        if ((uint64_t)d < range_n_) return false;
        if ((uint64_t)d == range_n_) {
            ++range_n_;
            cache_.reset();
            return true;
        }
        to_bitmap();
    }
    if (contains(d)) return false;
    std::vector<uint64_t>& w = own();
    const size_t wi = d >> 6;
    if (wi >= w.size()) w.resize(wi + 1, 0);
    w[wi] |= 1ull << (d & 63);
    if (dense_->len >= 0) ++dense_->len;
    return true;
}
bool DocSet::remove(DocumentId d) {
    if (kind_ == Array) {
        auto it = std::lower_bound(ids_.begin(), ids_.end(), d);
        if (it == ids_.end() || *it != d) return false;
        ids_.erase(it);
        cache_.reset();
        return true;
    }
    if (!contains(d)) return false;
    if (kind_ == Range) {
        if ((uint64_t)d + 1 == range_n_) {
            if (--range_n_ == 0) kind_ = Array;
            cache_.reset();
            return true;
        }
        to_bitmap();
    }
    own()[d >> 6] &= ~(1ull << (d & 63));
    if (dense_->len >= 0) --dense_->len;
    return true;
}
uint64_t DocSet::len() const {
    if (kind_ == Array) return ids_.size();
    if (kind_ == Range) return range_n_;
    if (!dense_) return 0;
    int64_t n = dense_->len;
    if (n < 0) dense_->len = n = (int64_t)count_bits(dense_->w);
    return (uint64_t)n;
}
bool DocSet::is_empty() const {
    if (kind_ == Array) return ids_.empty();
    if (kind_ == Range) return range_n_ == 0;
    if (!dense_) return true;
    if (dense_->len >= 0) return dense_->len == 0;
    for (uint64_t w : dense_->w)
        if (w) return false;
    return true;
}
DocSet DocSet::intersect(const DocSet& o, uint64_t* len) const {
    DocSet r;
    if (kind_ == Array || o.kind_ == Array) {  // walk the array, probe the other
        const DocSet& a = kind_ == Array ? *this : o;
        const DocSet& b = kind_ == Array ? o : *this;
        for (DocumentId d : a.ids_)
            if (b.contains(d)) r.ids_.push_back(d);
        if (len) *len = r.ids_.size();
        return r;
    }
    if (kind_ == Range && o.kind_ == Range) {
        r = from_range(std::min(range_n_, o.range_n_));
        if (len) *len = r.range_n_;
        return r;
    }
    if (kind_ == Range || o.kind_ == Range) {
        const uint64_t n = kind_ == Range ? range_n_ : o.range_n_;
        const DocSet& b = kind_ == Range ? o : *this;
        const size_t have = b.dense_ ? b.dense_->w.size() : 0;
        if (n >= (uint64_t)have * 64) {  // the range covers the bitmap: the result IS the bitmap
            r = b;
            if (len) *len = r.len();
            return r;
        }
        std::vector<uint64_t> w(b.dense_->w.begin(), b.dense_->w.begin() + (size_t)((n + 63) / 64));
        if (n & 63) w.back() &= (1ull << (n & 63)) - 1;
        r = from_words(std::move(w));
        if (len) *len = r.len();
        return r;
    }
    r.kind_ = Bitmap;
    const std::vector<uint64_t>& x = words();
    const std::vector<uint64_t>& y = o.words();
    r.dense_ = std::make_shared<Block>(std::vector<uint64_t>(std::min(x.size(), y.size())));
    uint64_t n = 0;
    for (size_t i = 0; i < r.dense_->w.size(); ++i) {
        const uint64_t w = x[i] & y[i];
        r.dense_->w[i] = w;
        n += popcount64(w);
    }
    r.dense_->len = (int64_t)n;
    if (len) *len = n;
    return r;
}
DocSet DocSet::operator&(const DocSet& o) const { return intersect(o, nullptr); }
DocSet DocSet::operator|(const DocSet& o) const {
    DocSet r = *this;
    r |= o;
    return r;
}
DocSet& DocSet::operator|=(const DocSet& o) {
    if (o.kind_ == Array) {
        for (DocumentId d : o.ids_) insert(d);
        return *this;
    }
    if (kind_ == Range && o.kind_ == Range) {
        range_n_ = std::max(range_n_, o.range_n_);
        cache_.reset();
        return *this;
    }
    to_bitmap();
    std::vector<uint64_t>& w = own();
    const std::vector<uint64_t>& y = o.words();
    if (y.size() > w.size()) w.resize(y.size(), 0);
    for (size_t i = 0; i < y.size(); ++i) w[i] |= y[i];
    dense_->len = -1;
    return *this;
}
DocSet& DocSet::operator-=(const DocSet& o) {
    if (kind_ == Array) {
        ids_.erase(std::remove_if(ids_.begin(), ids_.end(), [&](DocumentId d) { return o.contains(d); }), ids_.end());
        cache_.reset();
        return *this;
    }
    if (o.kind_ == Array) {
        for (DocumentId d : o.ids_) remove(d);
        return *this;
    }
    if (o.is_empty()) return *this;
    to_bitmap();
    std::vector<uint64_t>& w = own();
    const std::vector<uint64_t>& y = o.words();
    const size_t n = std::min(w.size(), y.size());
    for (size_t i = 0; i < n; ++i) w[i] &= ~y[i];
    dense_->len = -1;
    return *this;
}
std::vector<DocumentId> DocSet::to_ids() const {
    if (kind_ == Array) return ids_;
    std::vector<DocumentId> ids;
    if (kind_ == Range) {
        ids.resize((size_t)range_n_);
        for (uint64_t i = 0; i < range_n_; ++i) ids[(size_t)i] = (DocumentId)i;
        return ids;
    }
    const std::vector<uint64_t>& x = words();
    for (size_t i = 0; i < x.size(); ++i) {
        uint64_t w = x[i];
        while (w) {
            ids.push_back((DocumentId)(i * 64 + (size_t)__builtin_ctzll(w)));
            w &= w - 1;
        }
    }
    return ids;
}

// ---- DistributionShift (src/vector/mod.rs:395-433) ----------------------------------------

std::optional<DistributionShift> DistributionShift::create(float mean, float sigma) {
    if (sigma <= 0.0f) return std::nullopt;
    return DistributionShift{mean, sigma};
}

float DistributionShift::shift(float score) const {
    // every operation in f32 and in the reference's order; no fused multiply-add
    const volatile float target_mean = 0.5f;
    const volatile float target_sigma = 0.4f;
    const volatile float factor = target_sigma / current_sigma;
    const volatile float scaled_mean = factor * current_mean;
    const volatile float offset = target_mean - scaled_mean;
    const volatile float scaled = factor * score;
    float out = scaled + offset;
    // clamp the final score in the ]0, 1] interval
    if (out <= 0.0f) out = std::numeric_limits<float>::epsilon();
    if (out > 1.0f) out = 1.0f;
    return out;
}

// ---- score details (src/score_details.rs) ------------------------------------------------------

Rank Rank::merge(Rank outer, Rank inner) {
    outer.rank = outer.rank ? outer.rank - 1 : 0;  // saturating_sub(1)
    outer.rank *= inner.max_rank;
    outer.max_rank *= inner.max_rank;
    outer.rank += inner.rank;
    return outer;
}

double Rank::global_score(const std::vector<Rank>& details) {
    Rank rank{1, 1};
    for (const Rank& inner : details) rank = merge(rank, inner);
    return rank.local_score();
}

std::optional<int> Sort::partial_cmp(const Sort& other) const {
    if (field_name != other.field_name) return std::nullopt;
    if (ascending != other.ascending) return std::nullopt;
    const SortValue &l = value, &r = other.value;
    if (l.kind == SortValue::Null && r.kind == SortValue::Null) return 0;
    if (l.kind == SortValue::Null) return -1;
    if (r.kind == SortValue::Null) return 1;
    // numbers are always before strings
    if (l.kind == SortValue::Number && r.kind == SortValue::String) return 1;
    if (l.kind == SortValue::String && r.kind == SortValue::Number) return -1;
    if (l.kind == SortValue::Number && r.kind == SortValue::Number) {
        if (std::isnan(l.number) || std::isnan(r.number)) return std::nullopt;
        const int order = l.number < r.number ? -1 : l.number > r.number ? 1 : 0;
        // when ascending, smaller is better
        return ascending ? -order : order;
    }
    if (l.kind == SortValue::String && r.kind == SortValue::String) {
        const int c = l.string.compare(r.string);
        const int order = c < 0 ? -1 : c > 0 ? 1 : 0;
        return ascending ? -order : order;
    }
    return std::nullopt;
}

ScoreDetails ScoreDetails::vector(std::optional<float> similarity) {
    ScoreDetails d;
    d.kind = Vector;
    d.similarity = similarity;
    return d;
}
ScoreDetails ScoreDetails::ranked(Kind kind, uint32_t value, uint32_t max_value) {
    ScoreDetails d;
    d.kind = kind;
    d.value = value;
    d.max_value = max_value;
    return d;
}
ScoreDetails ScoreDetails::sorted(Sort sort) {
    ScoreDetails d;
    d.kind = SortKind;
    d.sort = std::move(sort);
    return d;
}

std::optional<Rank> ScoreDetails::rank() const {
    switch (kind) {
        case Words:      // score_details.rs:183-185
        case Exactness:  // :203-206
        case Attribute:  // :269-271
            return Rank{value + 1, max_value + 1};
        case Typo:       // :247-252
        case Proximity:  // :225-230
        {
            const uint32_t top = max_value + 1;
            return Rank{top > value ? top - value : 0, top};
        }
        default:
            return std::nullopt;
    }
}

std::vector<ScoreValue> score_values(const std::vector<ScoreDetails>& details) {
    // rank_or_value, then coalesce runs of ranks with Rank::merge (score_details.rs:59-89)
    std::vector<ScoreValue> out;
    std::optional<Rank> pending;
    auto flush = [&]() {
        if (pending) {
            ScoreValue v;
            v.score = pending->local_score();
            out.push_back(v);
            pending.reset();
        }
    };
    for (const ScoreDetails& d : details) {
        if (std::optional<Rank> r = d.rank()) {
            pending = pending ? Rank::merge(*pending, *r) : *r;
        } else if (d.kind == ScoreDetails::SortKind) {
            flush();
            ScoreValue v;
            v.is_score = false;
            v.sort = &d.sort;
            out.push_back(v);
        } else {
            flush();
            ScoreValue v;
            v.score = d.similarity ? (double)*d.similarity : 0.0;
            out.push_back(v);
        }
    }
    flush();
    return out;
}

double global_score(const std::vector<ScoreDetails>& details) {
    for (const ScoreValue& v : score_values(details))
        if (v.is_score) return v.score;
    return 1.0;
}

namespace {
std::string fmt_f64(double v) {
    if (!std::isfinite(v)) return "null";  // serde_json renders non-finite floats as null
    char buf[64];
    auto res = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, res.ptr);
    if (s.find_first_of(".en") == std::string::npos) s += ".0";
    return s;
}
std::string json_string(const std::string& s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        if (c == '"') o += "\\\"";
        else if (c == '\\') o += "\\\\";
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c == '\r') o += "\\r";
        else if (c < 0x20) {
            char b[8];
            snprintf(b, sizeof b, "\\u%04x", c);
            o += b;
        } else o += (char)c;
    }
    return o + "\"";
}
std::string sort_value_json(const SortValue& v) {
    switch (v.kind) {
        case SortValue::Number: return fmt_f64(v.number);
        case SortValue::String: return json_string(v.string);
        default: return "null";
    }
}
}  // namespace

std::string to_json_map(const std::vector<ScoreDetails>& details) {
    // score_details.rs:91-171; serde_json is built with preserve_order, so keys
    // keep insertion order, and a repeated key overwrites in place.
    std::vector<std::pair<std::string, std::string>> entries;
    auto put = [&](const std::string& key, const std::string& value) {
        for (auto& e : entries)
            if (e.first == key) {
                e.second = value;
                return;
            }
        entries.emplace_back(key, value);
    };
    int order = 0;
    for (const ScoreDetails& d : details) {
        const std::string o = std::to_string(order);
        auto ranked = [&](const char* name, const char* a, const char* b) {
            put(name, "{\"order\":" + o + ",\"" + a + "\":" + std::to_string(d.value) + ",\"" + b +
                          "\":" + std::to_string(d.max_value) + ",\"score\":" +
                          fmt_f64(d.rank()->local_score()) + "}");
        };
        switch (d.kind) {
            case ScoreDetails::Words: ranked("words", "matchingWords", "maxMatchingWords"); break;
            case ScoreDetails::Typo: ranked("typo", "typoCount", "maxTypoCount"); break;
            case ScoreDetails::Proximity: ranked("proximity", "proximity", "maxProximity"); break;
            case ScoreDetails::Attribute: ranked("attribute", "attribute", "maxAttribute"); break;
            case ScoreDetails::Exactness: ranked("exactness", "exactWords", "maxExactWords"); break;
            case ScoreDetails::SortKind:
                put(d.sort.field_name + (d.sort.ascending ? ":asc" : ":desc"),
                    "{\"order\":" + o + ",\"value\":" + sort_value_json(d.sort.value) + "}");
                break;
            case ScoreDetails::Vector:
                put("vectorSort", "{\"order\":" + o + ",\"similarity\":" +
                                      (d.similarity ? fmt_f64((double)*d.similarity) : "null") + "}");
                break;
        }
        ++order;
    }
    std::string out = "{";
    for (size_t i = 0; i < entries.size(); ++i) {
        if (i) out += ",";
        out += json_string(entries[i].first) + ":" + entries[i].second;
    }
    return out + "}";
}

// ---- bucket_sort (src/search/ranking/bucket_sort.rs:17-84) ------------------------------------------

namespace {
void recursive_sort(size_t current_rule, std::vector<ScoreDetails>& score,
                    std::vector<std::unique_ptr<RankingRule>>& ranking_rule, uint64_t& limit,
                    uint64_t& skip, std::vector<std::pair<DocumentId, std::vector<ScoreDetails>>>& output,
                    DocSet& visited) {
    for (;;) {
        // The reference asks its rule for one more bucket before it notices that the limit
        // is used up (bucket_sort.rs:45-49) and drops it. With VectorSort that extra bucket
        // is a cache refill, i.e. a second full nns_by_vector per query; asking only while
        // something can still be returned gives the same output for half the scans.
        if (limit == 0) return;
        std::optional<RankingRuleOutput> bucket = ranking_rule[current_rule]->next_bucket();
        if (!bucket) break;
        DocSet& docids = bucket->candidates;
        docids -= visited;
        const uint64_t len = docids.len();
        if (len == 0) continue;
        if (skip > 0 && len < skip) {
            visited |= docids;
            skip -= len;
            continue;
        }
        score.push_back(bucket->score);
        if (current_rule == ranking_rule.size() - 1) {
            visited |= docids;
            const uint64_t docids_count = len;
            uint64_t seen = 0, taken = 0;
            for (DocumentId d : docids.to_ids()) {
                if (seen++ < skip) continue;
                if (taken++ >= limit) break;
                output.emplace_back(d, score);
            }
            const uint64_t used = docids_count < skip ? 0 : docids_count - skip;
            limit = limit > used ? limit - used : 0;  // saturating_sub
            skip = docids_count < skip ? skip - docids_count : 0;
            score.pop_back();
        } else {
            ranking_rule[current_rule + 1]->start_iteration(bucket->candidates);
            recursive_sort(current_rule + 1, score, ranking_rule, limit, skip, output, visited);
        }
    }
    if (!score.empty()) score.pop_back();  // Vec::pop on an empty vector is a no-op
}
}  // namespace

BucketSortOutput bucket_sort(std::vector<std::unique_ptr<RankingRule>>& ranking_rules, uint64_t limit,
                             uint64_t skip, DocSet candidates) {
    BucketSortOutput out;
    if (ranking_rules.empty()) {
        out.docids = candidates.to_ids();
        out.scores.assign(out.docids.size(), {});
        out.candidates = std::move(candidates);
        return out;
    }
    DocSet visited;
    std::vector<ScoreDetails> score;
    std::vector<std::pair<DocumentId, std::vector<ScoreDetails>>> output;
    ranking_rules[0]->start_iteration(std::move(candidates));
    recursive_sort(0, score, ranking_rules, limit, skip, output, visited);
    for (auto& e : output) {
        out.docids.push_back(e.first);
        out.scores.push_back(std::move(e.second));
    }
    out.candidates = DocSet::from_ids(out.docids);  // only the returned documents (bucket_sort.rs:39)
    return out;
}

// ---- GpuContext: the seam over the C ABI ----------------------------------------------------------------

namespace {
std::string last_error(vs_ctx* ctx) {
    char buf[512];
    vs_last_error(ctx, buf, sizeof buf);
    return buf;
}

// What a GpuReader leaves with a candidate set's words: how often it has searched under them and,
// from the second search on, their copy in device memory. A set that is searched once (a request's
// universe that never comes back) is sent along with its search; one that comes back (a refill of
// the same request, the next request under the same facet filter) is uploaded once more, into a
// vs_mask, and every later search names that mask instead of sending N / 8 bytes again.
// The slot is shared by every copy of a universe's words and by every thread searching under them;
// a search holds its own reference to the mask for as long as it runs, so a thread that finds the
// mask outdated (the store was given more room) and replaces it never frees words another thread's
// scan is still reading.
struct DeviceMaskSlot {
    std::mutex mu;
    uint32_t searches = 0;
    std::shared_ptr<vs_mask> mask;
};

// The device copy of `candidates` for `store`, or null: send the words with the search. Stores of
// one context never share an address while both live, so the store pointer names the slot; a mask
// that no longer fits its store (the store was given more room, or its address has been reused) is
// dropped and made again.
std::shared_ptr<vs_mask> kept_mask(vs_store* store, const DocSet& candidates) {
    std::shared_ptr<void> held = candidates.attachment(store);
    if (!held) {
        held = std::make_shared<DeviceMaskSlot>();
        candidates.attach(store, held);
    }
    DeviceMaskSlot& slot = *static_cast<DeviceMaskSlot*>(held.get());
    std::lock_guard<std::mutex> g(slot.mu);
    if (slot.mask && !vs_mask_fits(store, slot.mask.get())) slot.mask.reset();
    if (!slot.mask && slot.searches++ >= 1) {
        static const uint64_t none = 0;
        const std::vector<uint64_t>& w = candidates.words();
        vs_mask* made = nullptr;
        // (no room on the device: the plain search reports what is wrong)
        if (vs_mask_create(store, w.empty() ? &none : w.data(), w.size(), &made) == VS_OK)
            slot.mask = std::shared_ptr<vs_mask>(made, [](vs_mask* m) { vs_mask_destroy(m); });
    }
    return slot.mask;
}

void check_vs(vs_ctx* ctx, int32_t rc) {
    if (rc == VS_E_DIM_MISMATCH) {
        uint32_t e = 0, r = 0;
        vs_last_dim_mismatch(ctx, &e, &r);
        throw Error(Error::InvalidVectorDimensions,
                    "Invalid vector dimensions: expected: `" + std::to_string(e) + "`, found: `" + std::to_string(r) +
                        "`.",
                    e, r);
    }
    if (rc != VS_OK) throw Error(Error::ArroyError, last_error(ctx));
}

class GpuReader : public Reader {
   public:
    GpuReader(vs_ctx* ctx, vs_store* store) : ctx_(ctx), store_(store) {}
    std::vector<std::pair<DocumentId, float>> nns_by_vector(const std::vector<float>& target, size_t count,
                                                            const DocSet& candidates) override {
        if (count > VS_MAX_K)
            throw Error(Error::ArroyError, "limit " + std::to_string(count) + " exceeds VS_MAX_K");
        std::vector<uint32_t> ids(count ? count : 1);
        std::vector<float> dist(count ? count : 1);
        uint32_t n = 0;
        int32_t rc;
        if (const std::shared_ptr<vs_mask> mask = kept_mask(store_, candidates)) {
            rc = vs_search_mask(store_, target.data(), (uint32_t)target.size(), (uint32_t)count, mask.get(), ids.data(),
                                dist.data(), &n);
        } else {
            static const uint64_t none = 0;
            const std::vector<uint64_t>& w = candidates.words();
            // an empty candidate set is an empty mask, not "no filter"
            rc = vs_search(store_, target.data(), (uint32_t)target.size(), (uint32_t)count,
                           w.empty() ? &none : w.data(), w.size(), ids.data(), dist.data(), &n);
        }
        check(rc);
        std::vector<std::pair<DocumentId, float>> out(n);
        for (uint32_t i = 0; i < n; ++i) out[i] = {ids[i], dist[i]};
        return out;
    }

    std::vector<std::vector<std::pair<DocumentId, float>>> nns_by_vectors(
        const std::vector<std::vector<float>>& targets, size_t count, const DocSet& candidates) override {
        std::vector<std::vector<std::pair<DocumentId, float>>> out(targets.size());
        if (targets.empty()) return out;
        if (count > VS_MAX_K)
            throw Error(Error::ArroyError, "limit " + std::to_string(count) + " exceeds VS_MAX_K");
        const size_t dim = targets[0].size(), n_q = targets.size(), kk = count ? count : 1;
        std::vector<float> flat(n_q * dim);
        for (size_t i = 0; i < n_q; ++i) {
            if (targets[i].size() != dim)  // one call needs one dimension; let the single path report it
                return Reader::nns_by_vectors(targets, count, candidates);
            std::copy(targets[i].begin(), targets[i].end(), flat.begin() + i * dim);
        }
        std::vector<uint32_t> ids(n_q * kk), n(n_q, 0);
        std::vector<float> dist(n_q * kk);
        int32_t rc;
        if (const std::shared_ptr<vs_mask> held = kept_mask(store_, candidates)) {
            const std::vector<const vs_mask*> masks(n_q, held.get());  // one filter: the queries share passes
            rc = vs_search_batch_masks(store_, flat.data(), (uint32_t)n_q, (uint32_t)dim, (uint32_t)count, nullptr,
                                       nullptr, masks.data(), ids.data(), dist.data(), n.data());
        } else {
            static const uint64_t none = 0;
            const std::vector<uint64_t>& w = candidates.words();
            rc = vs_search_batch(store_, flat.data(), (uint32_t)n_q, (uint32_t)dim, (uint32_t)count,
                                 w.empty() ? &none : w.data(), w.size(), ids.data(), dist.data(), n.data());
        }
        check(rc);
        for (size_t i = 0; i < n_q; ++i)
            for (uint32_t j = 0; j < n[i]; ++j) out[i].emplace_back(ids[i * count + j], dist[i * count + j]);
        return out;
    }

    std::vector<std::vector<std::pair<DocumentId, float>>> nns_by_vectors_each(
        const std::vector<std::vector<float>>& targets, size_t count,
        const std::vector<const DocSet*>& candidates) override {
        std::vector<std::vector<std::pair<DocumentId, float>>> out(targets.size());
        if (targets.empty()) return out;
        if (candidates.size() != targets.size())
            throw std::logic_error("nns_by_vectors_each: one candidate set per target");
        if (count > VS_MAX_K)
            throw Error(Error::ArroyError, "limit " + std::to_string(count) + " exceeds VS_MAX_K");
        const size_t dim = targets[0].size(), n_q = targets.size(), kk = count ? count : 1;
        std::vector<float> flat(n_q * dim);
        for (size_t i = 0; i < n_q; ++i) {
            if (targets[i].size() != dim) return Reader::nns_by_vectors_each(targets, count, candidates);
            std::copy(targets[i].begin(), targets[i].end(), flat.begin() + i * dim);
        }
        // per query: the device copy of its set if one is kept, else the set's words. Queries whose sets
        // share their words (copies of one universe object) name the same pointer and share passes.
        static const uint64_t none = 0;
        std::vector<std::shared_ptr<vs_mask>> held(n_q);  // (kept alive until the call has returned)
        std::vector<const vs_mask*> masks(n_q, nullptr);
        std::vector<const uint64_t*> words(n_q, nullptr);
        std::vector<uint64_t> n_words(n_q, 0);
        for (size_t i = 0; i < n_q; ++i) {
            held[i] = kept_mask(store_, *candidates[i]);
            masks[i] = held[i].get();
            const std::vector<uint64_t>& w = candidates[i]->words();
            words[i] = w.empty() ? &none : w.data();  // an empty set is an empty mask, not "no filter"
            n_words[i] = w.empty() ? 1 : w.size();
        }
        std::vector<uint32_t> ids(n_q * kk), n(n_q, 0);
        std::vector<float> dist(n_q * kk);
        check(vs_search_batch_masks(store_, flat.data(), (uint32_t)n_q, (uint32_t)dim, (uint32_t)count, words.data(),
                                    n_words.data(), masks.data(), ids.data(), dist.data(), n.data()));
        for (size_t i = 0; i < n_q; ++i)
            for (uint32_t j = 0; j < n[i]; ++j) out[i].emplace_back(ids[i * count + j], dist[i * count + j]);
        return out;
    }

   private:
    void check(int32_t rc) { check_vs(ctx_, rc); }
    vs_ctx* ctx_;
    vs_store* store_;
};
}  // namespace

std::vector<std::vector<std::pair<DocumentId, float>>> Reader::nns_by_vectors(
    const std::vector<std::vector<float>>& targets, size_t count, const DocSet& candidates) {
    std::vector<std::vector<std::pair<DocumentId, float>>> out;
    for (const auto& t : targets) out.push_back(nns_by_vector(t, count, candidates));
    return out;
}

std::vector<std::vector<std::pair<DocumentId, float>>> Reader::nns_by_vectors_each(
    const std::vector<std::vector<float>>& targets, size_t count, const std::vector<const DocSet*>& candidates) {
    std::vector<std::vector<std::pair<DocumentId, float>>> out;
    for (size_t i = 0; i < targets.size(); ++i) out.push_back(nns_by_vector(targets[i], count, *candidates.at(i)));
    return out;
}

uint8_t GpuContext::embedder_category_id(const std::string& name) {
    auto it = ids_.find(name);
    if (it == ids_.end()) throw Error(Error::InvalidEmbedder, "Invalid embedder `" + name + "`.");
    return it->second;
}

std::unique_ptr<Reader> GpuContext::vector_reader(uint16_t index) {
    vs_store* store = nullptr;
    const int32_t rc = vs_store_open(ctx_, index, &store);
    if (rc == VS_E_NO_STORE) return nullptr;  // MissingMetadata
    if (rc != VS_OK) throw Error(Error::ArroyError, last_error(ctx_));
    return std::unique_ptr<Reader>(new GpuReader(ctx_, store));
}

std::optional<std::vector<std::vector<std::pair<DocumentId, float>>>> GpuContext::nns_all_stores(
    uint8_t embedder_index, const std::vector<float>& target, size_t count, const DocSet& candidates) {
    static const bool reader_by_reader = getenv("VS_MIRROR_STORE_LOOP") != nullptr;  // measurement switch
    if (reader_by_reader) return std::nullopt;
    std::vector<std::vector<std::pair<DocumentId, float>>> per_store;
    const uint16_t base = (uint16_t)((uint16_t)embedder_index << 8);
    vs_store* first = nullptr;
    const int32_t opened = vs_store_open(ctx_, base, &first);
    if (opened == VS_E_NO_STORE) return per_store;  // MissingMetadata for k = 0: no reader, no result
    check_vs(ctx_, opened);
    if (count > VS_MAX_K) throw Error(Error::ArroyError, "limit " + std::to_string(count) + " exceeds VS_MAX_K");
    // the universe's device copy, kept under the first store: it serves every store that starts at the
    // same docid and is no larger; otherwise (and the first time a universe is seen) the words go along
    const std::shared_ptr<vs_mask> held = kept_mask(first, candidates);
    static const uint64_t none = 0;
    const std::vector<uint64_t>& w = candidates.words();
    uint32_t n = 0, n_stores = 0;
    std::vector<uint32_t> ids;
    std::vector<float> dist;
    std::vector<uint8_t> from;
    auto search = [&](const vs_mask* mask, size_t cap) {
        ids.resize(cap ? cap : 1);
        dist.resize(cap ? cap : 1);
        from.resize(cap ? cap : 1);
        // an empty candidate set is an empty mask, not "no filter"
        return vs_search_stores(ctx_, base, target.data(), (uint32_t)target.size(), (uint32_t)count,
                                mask ? nullptr : (w.empty() ? &none : w.data()), mask ? 0 : w.size(), nullptr, 0, mask,
                                ids.data(), dist.data(), from.data(), (uint32_t)cap, &n, &n_stores);
    };
    const vs_mask* mask = held.get();
    int32_t rc = search(mask, count * 4);
    if (rc == VS_E_INVALID_ARG && mask) {  // the mask does not cover some store: send the words
        mask = nullptr;
        rc = search(mask, count * 4);
    }
    if (rc == VS_E_CAPACITY) rc = search(mask, count * (size_t)n_stores);  // more than four stores
    check_vs(ctx_, rc);
    per_store.resize(n_stores);
    for (uint32_t i = 0; i < n; ++i) per_store[from[i]].emplace_back(ids[i], dist[i]);
    return per_store;
}

// ---- VectorSort (src/search/ranking/vector_sort.rs) ------------------------------------------------------

VectorSort::VectorSort(Context& ctx, std::vector<float> target, DocSet vector_candidates, size_t limit,
                       const std::string& embedder_name, std::optional<DistributionShift> distribution)
    : ctx_(ctx),
      target_(std::move(target)),
      vector_candidates_(std::move(vector_candidates)),
      limit_(limit),
      distribution_shift_(distribution),
      embedder_index_(ctx.embedder_category_id(embedder_name)) {}

std::vector<std::unique_ptr<Reader>> VectorSort::open_readers(Context& ctx, uint8_t embedder_index) {
    const uint16_t writer_index = (uint16_t)((uint16_t)embedder_index << 8);
    std::vector<std::unique_ptr<Reader>> readers;
    for (uint32_t k = 0; k <= 0xFF; ++k) {
        std::unique_ptr<Reader> reader = ctx.vector_reader((uint16_t)(writer_index | k));
        if (!reader) break;  // MissingMetadata ends the map_while
        readers.push_back(std::move(reader));
    }
    return readers;
}

void VectorSort::set_cache(std::vector<std::vector<std::pair<DocumentId, float>>> per_store_results) {
    std::vector<std::pair<DocumentId, float>> results;
    for (auto& nns : per_store_results) results.insert(results.end(), nns.begin(), nns.end());
    // The reference sorts by distance alone with an unstable sort, so the order
    // of equal distances from different stores is unspecified there; here it is
    // fixed as (distance, docid, store order), which a stable sort on
    // (distance, docid) gives. NaN sorts last, as OrderedFloat does.
    std::stable_sort(results.begin(), results.end(), [](const auto& a, const auto& b) {
        const bool an = std::isnan(a.second), bn = std::isnan(b.second);
        if (an != bn) return bn;
        if (!an && a.second != b.second) return a.second < b.second;
        return a.first < b.first;
    });
    cached_sorted_docids_ = std::move(results);
    cursor_ = 0;
}

void VectorSort::fill_buffer() {
    if (preloaded_) {  // the first fill was computed for this rule together with other queries'
        auto r = std::move(*preloaded_);
        preloaded_.reset();
        set_cache(std::move(r));
        return;
    }
    if (!emitted_.is_empty()) {  // a refill: the documents handed out so far leave the candidates for good
        live_ -= emitted_;
        candidates_ -= emitted_;
        emitted_ = DocSet();
        emitted_from_live_ = 0;
    }
    // one call for every store of the embedder where the context has one, else reader by reader
    if (auto all = ctx_.nns_all_stores(embedder_index_, target_, limit_, live_)) {
        set_cache(std::move(*all));
        return;
    }
    std::vector<std::vector<std::pair<DocumentId, float>>> per_store;
    for (auto& reader : open_readers(ctx_, embedder_index_))
        per_store.push_back(reader->nns_by_vector(target_, limit_, live_));
    set_cache(std::move(per_store));
}

void VectorSort::preload(std::vector<std::vector<std::pair<DocumentId, float>>> per_store_results) {
    preloaded_ = std::move(per_store_results);
}

void VectorSort::start_iteration(DocSet candidates) {
    candidates_ = std::move(candidates);
    live_ = vector_candidates_.intersect(candidates_, nullptr);
    emitted_ = DocSet();
    emitted_from_live_ = 0;
    finished_ = false;
    fill_buffer();
}

std::optional<RankingRuleOutput> VectorSort::next_bucket() {
    // The reference looks at `candidates` and at `vector_candidates & candidates` before it looks
    // at its cache (vector_sort.rs:96-109). If either is empty no cached document can still be
    // live, so looking at the cache first gives the same answers -- and the two cardinalities, a
    // pass over N / 64 words each for a universe nobody has counted yet, are only needed once the
    // cache has run dry, which a request that is served from its first fill never sees.
    for (;;) {  // the reference recurses after a refill
        if (finished_) return std::nullopt;
        while (cursor_ < cached_sorted_docids_.size()) {
            const auto [docid, distance] = cached_sorted_docids_[cursor_++];
            if (live_.contains(docid) && !emitted_.contains(docid)) {
                float score = 1.0f - distance;
                if (distribution_shift_) score = distribution_shift_->shift(score);
                emitted_.insert(docid);
                ++emitted_from_live_;
                return RankingRuleOutput{ScoreDetails::vector(score), DocSet::from_ids({docid})};
            }
        }
        // every emitted document was in live_, and live_ is a subset of the candidates
        if (candidates_.len() == emitted_from_live_) {  // `candidates` is empty (:96-98)
            finished_ = true;
            return std::nullopt;
        }
        if (live_.len() == emitted_from_live_) {
            // Mirrors the reference's assignment order (vector_sort.rs:102-109): the
            // candidates are cleared first, so the bucket that is returned is empty.
            finished_ = true;
            return RankingRuleOutput{ScoreDetails::vector(std::nullopt), DocSet()};
        }
        // the cache is exhausted: refill it and run again
        fill_buffer();
        if (cached_sorted_docids_.empty()) {
            // no document with a vector is left in the universe
            finished_ = true;
            return RankingRuleOutput{ScoreDetails::vector(std::nullopt), DocSet()};
        }
    }
}

// ---- execute_vector_search (src/search/search.rs:98-135, ranking_rule.rs:74-132) ----------------------------

SearchResult execute_vector_search(Context& ctx, const std::vector<float>& vector, DocSet universe,
                                   uint64_t skip, uint64_t limit, const std::string& embedder_name,
                                   std::optional<DistributionShift> distribution,
                                   const ExtraRules& extra_rules) {
    std::vector<std::unique_ptr<RankingRule>> ranking_rules;
    DocSet vector_candidates = ctx.all_docids();
    ranking_rules.emplace_back(new VectorSort(ctx, vector, std::move(vector_candidates),
                                              (size_t)(skip + limit), embedder_name, distribution));
    if (extra_rules)
        for (auto& r : extra_rules()) ranking_rules.push_back(std::move(r));
    BucketSortOutput out = bucket_sort(ranking_rules, limit, skip, std::move(universe));
    SearchResult res;
    res.candidates = std::move(out.candidates);
    res.documents_ids = std::move(out.docids);
    res.document_scores = std::move(out.scores);
    return res;
}

std::vector<SearchResult> execute_vector_search_batch(Context& ctx, const std::vector<std::vector<float>>& vectors,
                                                      DocSet universe, uint64_t skip, uint64_t limit,
                                                      const std::string& embedder_name,
                                                      std::optional<DistributionShift> distribution,
                                                      const ExtraRules& extra_rules) {
    // what every query's VectorSort asks its stores first: the nearest skip + limit items of
    // all_docids & universe (ranking_rule.rs:85, vector_sort.rs:90-93)
    const uint8_t embedder_index = ctx.embedder_category_id(embedder_name);
    const DocSet all = ctx.all_docids();
    const DocSet first_candidates = all & universe;
    std::vector<std::vector<std::vector<std::pair<DocumentId, float>>>> per_query(vectors.size());
    for (auto& reader : VectorSort::open_readers(ctx, embedder_index)) {
        auto r = reader->nns_by_vectors(vectors, (size_t)(skip + limit), first_candidates);
        for (size_t i = 0; i < vectors.size(); ++i) per_query[i].push_back(std::move(r[i]));
    }
    std::vector<SearchResult> out;
    for (size_t i = 0; i < vectors.size(); ++i) {
        std::vector<std::unique_ptr<RankingRule>> ranking_rules;
        auto* vs = new VectorSort(ctx, vectors[i], all, (size_t)(skip + limit), embedder_name, distribution);
        vs->preload(std::move(per_query[i]));
        ranking_rules.emplace_back(vs);
        if (extra_rules)
            for (auto& r : extra_rules()) ranking_rules.push_back(std::move(r));
        BucketSortOutput b = bucket_sort(ranking_rules, limit, skip, universe);
        SearchResult res;
        res.candidates = std::move(b.candidates);
        res.documents_ids = std::move(b.docids);
        res.document_scores = std::move(b.scores);
        out.push_back(std::move(res));
    }
    return out;
}

std::vector<SearchResult> execute_vector_search_batch(Context& ctx, const std::vector<std::vector<float>>& vectors,
                                                      const std::vector<DocSet>& universes, uint64_t skip,
                                                      uint64_t limit, const std::string& embedder_name,
                                                      std::optional<DistributionShift> distribution,
                                                      const ExtraRules& extra_rules) {
    if (universes.size() != vectors.size())
        throw std::logic_error("execute_vector_search_batch: one universe per vector");
    const uint8_t embedder_index = ctx.embedder_category_id(embedder_name);
    const DocSet all = ctx.all_docids();
    std::vector<DocSet> first(vectors.size());
    std::vector<const DocSet*> first_ptrs(vectors.size());
    for (size_t i = 0; i < vectors.size(); ++i) {
        first[i] = all & universes[i];
        first_ptrs[i] = &first[i];
    }
    std::vector<std::vector<std::vector<std::pair<DocumentId, float>>>> per_query(vectors.size());
    for (auto& reader : VectorSort::open_readers(ctx, embedder_index)) {
        auto r = reader->nns_by_vectors_each(vectors, (size_t)(skip + limit), first_ptrs);
        for (size_t i = 0; i < vectors.size(); ++i) per_query[i].push_back(std::move(r[i]));
    }
    std::vector<SearchResult> out;
    for (size_t i = 0; i < vectors.size(); ++i) {
        std::vector<std::unique_ptr<RankingRule>> ranking_rules;
        auto* vs = new VectorSort(ctx, vectors[i], all, (size_t)(skip + limit), embedder_name, distribution);
        vs->preload(std::move(per_query[i]));
        ranking_rules.emplace_back(vs);
        if (extra_rules)
            for (auto& r : extra_rules()) ranking_rules.push_back(std::move(r));
        BucketSortOutput b = bucket_sort(ranking_rules, limit, skip, universes[i]);
        SearchResult res;
        res.candidates = std::move(b.candidates);
        res.documents_ids = std::move(b.docids);
        res.document_scores = std::move(b.scores);
        out.push_back(std::move(res));
    }
    return out;
}

// ---- hybrid (src/search/hybrid.rs) ------------------------------------------------------------------------------

int compare_scores(const ScoreWithRatio& left, const ScoreWithRatio& right) {
    const std::vector<ScoreValue> l = score_values(left.first), r = score_values(right.first);
    const float left_ratio = left.second, right_ratio = right.second;
    for (size_t i = 0;; ++i) {
        const bool lh = i < l.size(), rh = i < r.size();
        if (!lh && !rh) return 0;
        if (!lh) return -1;
        if (!rh) return 1;
        if (l[i].is_score && r[i].is_score) {
            const double lv = l[i].score * (double)left_ratio;
            const double rv = r[i].score * (double)right_ratio;
            if (std::fabs(lv - rv) <= std::numeric_limits<double>::epsilon()) continue;
            return lv < rv ? -1 : 1;  // partial_cmp().unwrap(): scores are never NaN here
        }
        if (!l[i].is_score && !r[i].is_score) {
            const std::optional<int> c = l[i].sort->partial_cmp(*r[i].sort);
            if (!c) throw std::logic_error("compare_scores: incomparable sort values");  // unwrap()
            if (*c == 0) continue;
            return *c;
        }
        if (l[i].is_score) return l[i].score == 0.0 ? -1 : 1;
        return r[i].score == 0.0 ? 1 : -1;
    }
}

bool results_good_enough(const SearchResult& keyword_results, float semantic_ratio, uint64_t limit,
                         uint64_t offset) {
    // a result is good enough if its keyword score is > 0.9 with a semantic ratio of 0.5
    const double GOOD_ENOUGH_SCORE = 0.45;
    if ((uint64_t)keyword_results.document_scores.size() < limit + offset) return false;
    for (const auto& score : keyword_results.document_scores) {
        const double s = global_score(score);
        if (s * (double)(1.0f - semantic_ratio) < GOOD_ENOUGH_SCORE) return false;
    }
    return true;
}

std::pair<SearchResult, uint32_t> merge(const SearchResult& vector_results, float vector_ratio,
                                        const SearchResult& keyword_results, float keyword_ratio,
                                        uint64_t from, uint64_t length) {
    // itertools merge_by: take from the vector side while compare_scores(left, right).is_ge()
    SearchResult out;
    uint32_t semantic_hit_count = 0;
    DocSet documents_seen;
    size_t vi = 0, ki = 0;
    const size_t vn = vector_results.documents_ids.size(), kn = keyword_results.documents_ids.size();
    uint64_t skipped = 0;
    while (vi < vn || ki < kn) {
        bool take_vector;
        if (vi < vn && ki < kn) {
            take_vector = compare_scores({vector_results.document_scores[vi], vector_ratio},
                                         {keyword_results.document_scores[ki], keyword_ratio}) >= 0;
        } else {
            take_vector = vi < vn;
        }
        const SearchResult& src = take_vector ? vector_results : keyword_results;
        const size_t at = take_vector ? vi++ : ki++;
        const DocumentId docid = src.documents_ids[at];
        if (!documents_seen.insert(docid)) continue;  // remove documents we already saw
        if (skipped < from) {                         // start skipping after the filter
            ++skipped;
            continue;
        }
        if ((uint64_t)out.documents_ids.size() >= length) break;  // take after skipping
        if (take_vector) ++semantic_hit_count;
        out.documents_ids.push_back(docid);
        out.document_scores.push_back(src.document_scores[at]);
    }
    out.candidates = vector_results.candidates | keyword_results.candidates;
    return {std::move(out), semantic_hit_count};
}

std::pair<SearchResult, std::optional<uint32_t>> execute_hybrid(const HybridInput& in,
                                                                float semantic_ratio) {
    SearchResult keyword_results = in.keyword_search();
    // completely skip semantic search if the results of the keyword search are good enough
    if (results_good_enough(keyword_results, semantic_ratio, in.limit, in.offset))
        return {std::move(keyword_results), 0u};
    // no vector search against placeholder search
    if (!in.query) return {std::move(keyword_results), 0u};
    // no embedder, no semantic search
    if (!in.has_semantic) return {std::move(keyword_results), 0u};
    std::vector<float> vector_query;
    if (in.vector) {
        vector_query = *in.vector;
    } else {
        std::optional<std::vector<float>> e = in.embed_one ? in.embed_one(*in.query) : std::nullopt;
        if (!e) return {std::move(keyword_results), 0u};  // "Embedding failed": keyword results only
        vector_query = std::move(*e);
    }
    SearchResult vector_results = in.vector_search(vector_query);
    auto merged = merge(vector_results, semantic_ratio, keyword_results, 1.0f - semantic_ratio,
                        in.offset, in.limit);
    if ((uint64_t)merged.first.documents_ids.size() > in.limit)
        throw std::logic_error("execute_hybrid: more results than the limit");  // assert!
    return {std::move(merged.first), merged.second};
}

// ---- write path ------------------------------------------------------------------------------------------------

std::map<std::string, uint8_t> assign_embedder_category_ids(const std::vector<std::string>& embedder_names) {
    // new_configs is a BTreeMap<String, _>: iteration in byte-wise name order, duplicates collapse
    std::map<std::string, uint8_t> ids;
    for (const auto& name : embedder_names) ids.emplace(name, 0);
    size_t index = 0;
    for (auto& kv : ids) {
        if (index > 0xFF)  // `index.try_into()` into u8 (settings.rs:781-783)
            throw Error(Error::TooManyEmbedders, "Too many embedders in the configuration. Found " +
                                                     std::to_string(ids.size()) + ", but limited to 256.");
        kv.second = (uint8_t)index++;
    }
    return ids;
}

namespace {
// arroy::Writer over a vs_store. The indexer hands over items one at a time
// (typed_chunk.rs:618,650,668,691,701); sending each as an upload of its own costs a transfer and a
// stream wait per vector, so the writer keeps the pending changes on the host -- per docid its
// final state, an added row or a removal -- and hands them to the store in ONE vs_store_remove and
// ONE vs_store_upload when it is built (or when the pending rows pass 64 MiB). Reads of the writer's
// view are answered from the pending changes first: contains_item and item_vector see an item
// added a moment ago without anything being sent.
class GpuWriter : public Writer {
   public:
    GpuWriter(vs_ctx* ctx, uint16_t index, uint32_t dimension, uint64_t capacity)
        : ctx_(ctx), index_(index), dimension_(dimension), capacity_(capacity) {
        static const bool each = getenv("VS_MIRROR_WRITE_EACH") != nullptr;  // measurement switch: no batching
        flush_every_change_ = each;
    }
    ~GpuWriter() override = default;  // stores belong to the context

    bool del_item(DocumentId docid) override {
        if (!contains_item(docid)) return false;
        auto it = pending_.find(docid);
        if (it != pending_.end() && it->second != REMOVED) free_rows_.push_back(it->second);
        pending_[docid] = REMOVED;
        changed();
        return true;
    }
    void add_item(DocumentId docid, const std::vector<float>& vector) override {
        if (vector.size() != dimension_)
            throw Error(Error::InvalidVectorDimensions,
                        "Invalid vector dimensions: expected: `" + std::to_string(dimension_) + "`, found: `" +
                            std::to_string(vector.size()) + "`.",
                        dimension_, (uint32_t)vector.size());
        auto it = pending_.find(docid);
        size_t row;
        if (it != pending_.end() && it->second != REMOVED) {
            row = it->second;  // add_item twice: the last vector wins
        } else if (!free_rows_.empty()) {
            row = free_rows_.back();
            free_rows_.pop_back();
        } else {
            row = rows_.size() / dimension_;
            rows_.resize(rows_.size() + dimension_);
        }
        std::copy(vector.begin(), vector.end(), rows_.begin() + row * dimension_);
        pending_[docid] = row;
        changed();
    }
    std::optional<std::vector<float>> item_vector(DocumentId docid) override {
        auto it = pending_.find(docid);
        if (it != pending_.end()) {
            if (it->second == REMOVED) return std::nullopt;
            return std::vector<float>(rows_.begin() + it->second * dimension_,
                                      rows_.begin() + (it->second + 1) * dimension_);
        }
        if (!in_store(docid)) return std::nullopt;
        std::vector<float> out(dimension_);
        check(vs_store_item_vector(store_, docid, out.data()));
        return out;
    }
    bool contains_item(DocumentId docid) override {
        auto it = pending_.find(docid);
        if (it != pending_.end()) return it->second != REMOVED;
        return in_store(docid);
    }
    bool is_empty() override {
        flush();
        if (!attach()) return true;
        uint64_t n = 0;
        check(vs_store_writer_len(store_, &n));
        return n == 0;
    }
    void build() override {
        flush();
        if (attach()) check(vs_store_seal(store_));
    }
    // The reference does not rebuild a store whose last item was removed, and what arroy then
    // answers from the stale tree over deleted leaves cannot be established here. A slab must
    // not go on returning removed documents, so the emptied store is sealed.
    bool publish_if_emptied() override {
        flush();
        if (!attach()) return false;
        uint64_t sealed = 0;
        if (vs_store_len(store_, &sealed) == VS_OK && sealed != 0) check(vs_store_seal(store_));
        return true;
    }

   private:
    static constexpr size_t REMOVED = ~(size_t)0;
    // Hands the pending changes to the store: the removals in one call, the added rows in one call.
    void flush() {
        if (pending_.empty()) return;
        std::vector<uint32_t> gone, ids;
        std::vector<float> rows;
        DocumentId top = 0;
        for (const auto& kv : pending_) {
            if (kv.second == REMOVED) {
                gone.push_back(kv.first);
            } else {
                ids.push_back(kv.first);
                top = std::max(top, kv.first);
            }
        }
        // in docid order: neighbouring slots, and the same order whatever the hash map did
        std::sort(gone.begin(), gone.end());
        std::sort(ids.begin(), ids.end());
        rows.resize(ids.size() * dimension_);
        for (size_t i = 0; i < ids.size(); ++i) {
            const size_t row = pending_[ids[i]];
            std::copy(rows_.begin() + row * dimension_, rows_.begin() + (row + 1) * dimension_,
                      rows.begin() + i * dimension_);
        }
        if (!gone.empty() && attach()) check(vs_store_remove(store_, gone.data(), gone.size()));
        if (!ids.empty()) {
            if (!store_) {
                // an index built by an earlier run already has its store
                if (vs_store_open(ctx_, index_, &store_) != VS_OK) {
                    store_ = nullptr;
                    check(vs_store_create(ctx_, index_, dimension_, VS_F32, VS_ANGULAR, 0, capacity_, &store_));
                }
            }
            if ((uint64_t)top >= capacity_) {  // milli hands out ever larger docids
                while ((uint64_t)top >= capacity_) capacity_ *= 2;
                check(vs_store_reserve(store_, capacity_));
            }
            check(vs_store_upload(store_, ids.data(), rows.data(), ids.size()));
        }
        pending_.clear();
        rows_.clear();
        free_rows_.clear();
    }
    void changed() {
        if (flush_every_change_ || rows_.size() * sizeof(float) >= (64u << 20)) flush();
    }
    bool in_store(DocumentId docid) {
        if (!attach()) return false;
        int32_t has = 0;
        check(vs_store_contains(store_, docid, &has));
        return has != 0;
    }
    bool attach() {
        if (!store_ && vs_store_open(ctx_, index_, &store_) != VS_OK) store_ = nullptr;
        return store_ != nullptr;
    }
    void check(int32_t rc) {
        if (rc != VS_OK) throw Error(Error::ArroyError, last_error(ctx_));
    }
    vs_ctx* ctx_;
    uint16_t index_;
    uint32_t dimension_;
    uint64_t capacity_;
    vs_store* store_ = nullptr;
    bool flush_every_change_ = false;
    std::unordered_map<DocumentId, size_t> pending_;  // docid -> row of rows_, or REMOVED
    std::vector<float> rows_;                         // pending rows, dimension_ floats each
    std::vector<size_t> free_rows_;                   // rows of rows_ given up by a removal
};
}  // namespace

struct GpuWriters::Impl {
    std::vector<std::unique_ptr<GpuWriter>> writers;
};

GpuWriters::GpuWriters(vs_ctx* ctx, uint8_t embedder_index, uint32_t dimension, uint64_t capacity)
    : impl_(new Impl()) {
    const uint16_t writer_index = (uint16_t)((uint16_t)embedder_index << 8);
    for (uint32_t k = 0; k <= 0xFF; ++k)
        impl_->writers.emplace_back(new GpuWriter(ctx, (uint16_t)(writer_index | k), dimension, capacity));
}
GpuWriters::~GpuWriters() = default;
Writer& GpuWriters::writer(uint8_t k) { return *impl_->writers[k]; }

void write_vector_points(Writers& writers, const VectorPoints& points) {
    // remove vectors for docids we want them removed (typed_chunk.rs:610-622)
    for (DocumentId docid : points.remove_vectors)
        for (uint32_t k = 0; k <= 0xFF; ++k)
            // uses invariant: vectors are packed in the first writers
            if (!writers.writer((uint8_t)k).del_item(docid)) break;

    // add generated embeddings (:624-652)
    for (const auto& [docid, embeddings] : points.embeddings) {
        if (embeddings.size() > 0xFF)
            throw Error(Error::TooManyVectors, "Too many vectors for document with id " + std::to_string(docid) +
                                                   ": found " + std::to_string(embeddings.size()) +
                                                   ", but limited to 256.");
        for (size_t k = 0; k < embeddings.size(); ++k) writers.writer((uint8_t)k).add_item(docid, embeddings[k]);
    }

    // perform the manual diff (:654-707)
    for (const auto& diff : points.manual_vectors) {
        const DocumentId docid = diff.docid;
        if (diff.deletion) {
            std::optional<uint32_t> deleted_index;
            for (uint32_t index = 0; index <= 0xFF; ++index) {
                Writer& writer = writers.writer((uint8_t)index);
                const auto candidate = writer.item_vector(docid);
                if (!candidate) break;  // packed in the first writers
                if (*candidate == *diff.deletion) {
                    writer.del_item(docid);
                    deleted_index = index;
                }
            }
            // enforce the invariant: the last vector of the document moves into the hole
            if (deleted_index) {
                std::optional<std::pair<uint32_t, std::vector<float>>> last_index_with_a_vector;
                for (uint32_t index = *deleted_index; index <= 0xFF; ++index) {
                    auto candidate = writers.writer((uint8_t)index).item_vector(docid);
                    if (!candidate) break;
                    last_index_with_a_vector = std::make_pair(index, std::move(*candidate));
                }
                if (last_index_with_a_vector) {
                    writers.writer((uint8_t)last_index_with_a_vector->first).del_item(docid);
                    writers.writer((uint8_t)*deleted_index).add_item(docid, last_index_with_a_vector->second);
                }
            }
        }
        if (diff.addition) {
            // overflow was detected during vector extraction
            for (uint32_t k = 0; k <= 0xFF; ++k) {
                Writer& writer = writers.writer((uint8_t)k);
                if (!writer.contains_item(docid)) {
                    writer.add_item(docid, *diff.addition);
                    break;
                }
            }
        }
    }
}

uint32_t build_vector_stores(Writers& writers) {
    uint32_t built = 0;
    for (uint32_t k = 0; k <= 0xFF; ++k) {
        Writer& writer = writers.writer((uint8_t)k);
        if (writer.is_empty()) {
            // (the writers behind the first empty one hold nothing either: vectors are packed)
            for (uint32_t e = k; e <= 0xFF && writers.writer((uint8_t)e).publish_if_emptied(); ++e) {
            }
            break;
        }
        writer.build();
        ++built;
    }
    return built;
}

}  // namespace search_engine
