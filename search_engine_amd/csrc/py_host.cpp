// pybind11 binding of the C++ host mirror (host_engine.h) so that the parity
// tests can drive it from Python. Test plumbing only: a host application links
// host_engine.cpp (or keeps its own Rust code) and calls the C ABI directly.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <chrono>

#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "host_engine.h"

namespace py = pybind11;
using namespace search_engine;

namespace {

DocSet to_docset(const py::object& o) {
    if (o.is_none()) return DocSet();
    if (py::isinstance<py::array>(o)) {
        py::array a = py::reinterpret_borrow<py::array>(o);
        if (a.dtype().is(py::dtype::of<uint64_t>())) {
            auto w = py::array_t<uint64_t, py::array::c_style | py::array::forcecast>(a);
            return DocSet::from_words(std::vector<uint64_t>(w.data(), w.data() + w.size()));
        }
    }
    std::vector<DocumentId> ids;
    for (py::handle h : o) ids.push_back(h.cast<DocumentId>());
    return DocSet::from_ids(ids);
}

py::list from_docset(const DocSet& s) {
    py::list l;
    for (DocumentId d : s.to_ids()) l.append(d);
    return l;
}

const char* kind_name(ScoreDetails::Kind k) {
    switch (k) {
        case ScoreDetails::Words: return "words";
        case ScoreDetails::Typo: return "typo";
        case ScoreDetails::Proximity: return "proximity";
        case ScoreDetails::Exactness: return "exactness";
        case ScoreDetails::Attribute: return "attribute";
        case ScoreDetails::SortKind: return "sort";
        default: return "vector";
    }
}

py::object sort_value_to_py(const SortValue& v) {
    if (v.kind == SortValue::Number) return py::float_(v.number);
    if (v.kind == SortValue::String) return py::str(v.string);
    return py::none();
}

py::dict sort_to_py(const Sort& s) {
    py::dict d;
    d["kind"] = "sort";
    d["field"] = s.field_name;
    d["ascending"] = s.ascending;
    d["value"] = sort_value_to_py(s.value);
    return d;
}

py::dict details_to_py(const ScoreDetails& sd) {
    if (sd.kind == ScoreDetails::SortKind) return sort_to_py(sd.sort);
    py::dict d;
    d["kind"] = kind_name(sd.kind);
    if (sd.kind == ScoreDetails::Vector) {
        d["similarity"] = sd.similarity ? py::object(py::float_(*sd.similarity)) : py::object(py::none());
    } else {
        d["value"] = sd.value;
        d["max"] = sd.max_value;
    }
    return d;
}

ScoreDetails details_from_py(const py::handle& h) {
    py::dict d = py::reinterpret_borrow<py::dict>(h);
    const std::string kind = d["kind"].cast<std::string>();
    if (kind == "vector") {
        py::object s = d["similarity"];
        return ScoreDetails::vector(s.is_none() ? std::nullopt : std::optional<float>(s.cast<float>()));
    }
    if (kind == "sort") {
        Sort s;
        s.field_name = d["field"].cast<std::string>();
        s.ascending = d["ascending"].cast<bool>();
        py::object v = d["value"];
        if (v.is_none()) {
            s.value.kind = SortValue::Null;
        } else if (py::isinstance<py::str>(v)) {
            s.value.kind = SortValue::String;
            s.value.string = v.cast<std::string>();
        } else if (py::isinstance<py::float_>(v) || py::isinstance<py::int_>(v)) {
            s.value.kind = SortValue::Number;
            s.value.number = v.cast<double>();
        } else {
            s.value.kind = SortValue::Other;
        }
        return ScoreDetails::sorted(std::move(s));
    }
    ScoreDetails::Kind k = kind == "words"       ? ScoreDetails::Words
                           : kind == "typo"      ? ScoreDetails::Typo
                           : kind == "proximity" ? ScoreDetails::Proximity
                           : kind == "exactness" ? ScoreDetails::Exactness
                           : kind == "attribute" ? ScoreDetails::Attribute
                                                 : throw py::value_error("unknown score kind " + kind);
    return ScoreDetails::ranked(k, d["value"].cast<uint32_t>(), d["max"].cast<uint32_t>());
}

std::vector<ScoreDetails> details_list_from_py(const py::handle& h) {
    std::vector<ScoreDetails> v;
    for (py::handle e : h) v.push_back(details_from_py(e));
    return v;
}

py::list details_list_to_py(const std::vector<ScoreDetails>& v) {
    py::list l;
    for (const ScoreDetails& d : v) l.append(details_to_py(d));
    return l;
}

SearchResult result_from_py(const py::handle& h) {
    py::dict d = py::reinterpret_borrow<py::dict>(h);
    SearchResult r;
    r.documents_ids = d["documents_ids"].cast<std::vector<DocumentId>>();
    for (py::handle s : d["document_scores"]) r.document_scores.push_back(details_list_from_py(s));
    r.candidates = to_docset(d["candidates"]);
    return r;
}

py::dict result_to_py(const SearchResult& r) {
    py::dict d;
    d["documents_ids"] = r.documents_ids;
    py::list scores;
    for (const auto& s : r.document_scores) scores.append(details_list_to_py(s));
    d["document_scores"] = scores;
    d["candidates"] = from_docset(r.candidates);
    return d;
}

std::optional<DistributionShift> distribution_from_py(const py::object& o) {
    if (o.is_none()) return std::nullopt;
    auto t = o.cast<std::pair<float, float>>();
    return DistributionShift::create(t.first, t.second);
}

// ---- Python-implemented Reader / Context / RankingRule (the injectable fakes) ----------

struct PyReader : Reader {
    py::object impl;
    explicit PyReader(py::object o) : impl(std::move(o)) {}
    std::vector<std::pair<DocumentId, float>> nns_by_vector(const std::vector<float>& target, size_t count,
                                                            const DocSet& candidates) override {
        py::gil_scoped_acquire gil;
        py::object r = impl.attr("nns_by_vector")(target, count, from_docset(candidates));
        return r.cast<std::vector<std::pair<DocumentId, float>>>();
    }
};

// A Context whose three vector methods are the attributes of a Python object.
struct PyContext : Context {
    py::object impl;
    explicit PyContext(py::object o) : impl(std::move(o)) {}
    DocSet all_docids() override { return to_docset(impl.attr("all_docids")()); }
    uint8_t embedder_category_id(const std::string& name) override {
        py::object r = impl.attr("embedder_category_id")(name);
        if (r.is_none()) throw Error(Error::InvalidEmbedder, "Invalid embedder `" + name + "`.");
        return r.cast<uint8_t>();
    }
    std::unique_ptr<Reader> vector_reader(uint16_t index) override {
        py::object r = impl.attr("vector_reader")(index);
        if (r.is_none()) return nullptr;
        return std::unique_ptr<Reader>(new PyReader(r));
    }
};

struct PyRule : RankingRule {
    py::object impl;
    explicit PyRule(py::object o) : impl(std::move(o)) {}
    void start_iteration(DocSet candidates) override {
        impl.attr("start_iteration")(from_docset(candidates));
    }
    std::optional<RankingRuleOutput> next_bucket() override {
        py::object r = impl.attr("next_bucket")();
        if (r.is_none()) return std::nullopt;
        py::dict d = py::reinterpret_borrow<py::dict>(r);
        return RankingRuleOutput{details_from_py(d["score"]), to_docset(d["candidates"])};
    }
};

// forwards to a rule owned elsewhere (a VectorSort held by Python)
struct RuleRef : RankingRule {
    RankingRule* target;
    py::object keep_alive;
    RuleRef(RankingRule* t, py::object k) : target(t), keep_alive(std::move(k)) {}
    void start_iteration(DocSet candidates) override { target->start_iteration(std::move(candidates)); }
    std::optional<RankingRuleOutput> next_bucket() override { return target->next_bucket(); }
};

struct VectorSortHolder {
    std::shared_ptr<Context> ctx;
    std::unique_ptr<VectorSort> rule;
};

std::unique_ptr<RankingRule> rule_from_py(const py::handle& h) {
    py::object o = py::reinterpret_borrow<py::object>(h);
    if (py::isinstance<VectorSortHolder>(o))
        return std::unique_ptr<RankingRule>(new RuleRef(o.cast<VectorSortHolder&>().rule.get(), o));
    return std::unique_ptr<RankingRule>(new PyRule(o));
}

std::shared_ptr<Context> context_from_py(const py::object& o) {
    if (py::isinstance<GpuContext>(o)) return o.cast<std::shared_ptr<GpuContext>>();
    return std::make_shared<PyContext>(o);
}

// Writers whose 256 writers are Python objects (an in-memory fake of arroy::Writer for the CPU tests)
struct PyWriter : Writer {
    py::object impl;
    explicit PyWriter(py::object o) : impl(std::move(o)) {}
    bool del_item(DocumentId d) override { return impl.attr("del_item")(d).cast<bool>(); }
    void add_item(DocumentId d, const std::vector<float>& v) override { impl.attr("add_item")(d, v); }
    std::optional<std::vector<float>> item_vector(DocumentId d) override {
        py::object r = impl.attr("item_vector")(d);
        if (r.is_none()) return std::nullopt;
        return r.cast<std::vector<float>>();
    }
    bool contains_item(DocumentId d) override { return impl.attr("contains_item")(d).cast<bool>(); }
    bool is_empty() override { return impl.attr("is_empty")().cast<bool>(); }
    void build() override { impl.attr("build")(); }
};
struct PyWriters : Writers {
    py::object factory;  // factory(k) -> writer object
    std::map<uint8_t, std::unique_ptr<PyWriter>> made;
    explicit PyWriters(py::object f) : factory(std::move(f)) {}
    Writer& writer(uint8_t k) override {
        auto it = made.find(k);
        if (it == made.end()) it = made.emplace(k, std::unique_ptr<PyWriter>(new PyWriter(factory(k)))).first;
        return *it->second;
    }
};

VectorPoints points_from_py(const py::dict& d) {
    VectorPoints p;
    if (d.contains("remove_vectors")) p.remove_vectors = d["remove_vectors"].cast<std::vector<DocumentId>>();
    if (d.contains("embeddings"))
        for (py::handle e : d["embeddings"]) {
            auto t = e.cast<std::pair<DocumentId, std::vector<std::vector<float>>>>();
            p.embeddings.push_back(std::move(t));
        }
    if (d.contains("manual_vectors"))
        for (py::handle e : d["manual_vectors"]) {
            py::tuple t = py::reinterpret_borrow<py::tuple>(e);  // (docid, deletion | None, addition | None)
            VectorPoints::ManualDiff m;
            m.docid = t[0].cast<DocumentId>();
            if (!t[1].is_none()) m.deletion = t[1].cast<std::vector<float>>();
            if (!t[2].is_none()) m.addition = t[2].cast<std::vector<float>>();
            p.manual_vectors.push_back(std::move(m));
        }
    return p;
}

py::object bucket_to_py(const std::optional<RankingRuleOutput>& b) {
    if (!b) return py::none();
    py::dict d;
    d["score"] = details_to_py(b->score);
    d["candidates"] = from_docset(b->candidates);
    return d;
}

}  // namespace

PYBIND11_MODULE(_host, m) {
    m.doc() = "C++ mirror of the reference's host logic on the vector-search path";

    static py::exception<Error> host_error(m, "HostError");
    py::register_exception_translator([](std::exception_ptr p) {
        try {
            if (p) std::rethrow_exception(p);
        } catch (const Error& e) {
            const char* kind = e.kind == Error::InvalidEmbedder            ? "InvalidEmbedder"
                               : e.kind == Error::InvalidVectorDimensions ? "InvalidVectorDimensions"
                               : e.kind == Error::TooManyVectors          ? "TooManyVectors"
                               : e.kind == Error::TooManyEmbedders        ? "TooManyEmbedders"
                                                                          : "ArroyError";
            PyErr_SetObject(host_error.ptr(), py::make_tuple(e.what(), kind, e.expected, e.found).ptr());
        }
    });

    m.def("distribution_shift", [](float mean, float sigma, float score) -> py::object {
        auto d = DistributionShift::create(mean, sigma);
        if (!d) return py::none();
        return py::float_(d->shift(score));
    });
    m.def("rank_merge", [](std::pair<uint32_t, uint32_t> o, std::pair<uint32_t, uint32_t> i) {
        Rank r = Rank::merge(Rank{o.first, o.second}, Rank{i.first, i.second});
        return std::make_pair(r.rank, r.max_rank);
    });
    m.def("rank_global_score", [](const std::vector<std::pair<uint32_t, uint32_t>>& v) {
        std::vector<Rank> r;
        for (auto& p : v) r.push_back(Rank{p.first, p.second});
        return Rank::global_score(r);
    });
    m.def("score_values", [](const py::object& details) {
        const std::vector<ScoreDetails> d = details_list_from_py(details);
        py::list out;
        for (const ScoreValue& v : score_values(d)) {
            if (v.is_score) out.append(py::make_tuple("score", v.score));
            else out.append(py::make_tuple("sort", sort_to_py(*v.sort)));
        }
        return out;
    });
    m.def("global_score", [](const py::object& d) { return global_score(details_list_from_py(d)); });
    m.def("to_json_map", [](const py::object& d) { return to_json_map(details_list_from_py(d)); });
    m.def("sort_partial_cmp", [](const py::object& a, const py::object& b) -> py::object {
        auto c = details_from_py(a).sort.partial_cmp(details_from_py(b).sort);
        return c ? py::object(py::int_(*c)) : py::object(py::none());
    });
    m.def("compare_scores", [](const py::object& l, float lr, const py::object& r, float rr) {
        return compare_scores({details_list_from_py(l), lr}, {details_list_from_py(r), rr});
    });
    m.def("results_good_enough", [](const py::object& res, float ratio, uint64_t limit, uint64_t offset) {
        return results_good_enough(result_from_py(res), ratio, limit, offset);
    });
    m.def("merge", [](const py::object& v, float vr, const py::object& k, float kr, uint64_t from,
                      uint64_t length) {
        auto r = merge(result_from_py(v), vr, result_from_py(k), kr, from, length);
        return py::make_tuple(result_to_py(r.first), r.second);
    });
    m.def("execute_hybrid",
          [](py::object query, bool has_semantic, py::object vector, uint64_t offset, uint64_t limit,
             py::function keyword_search, py::function vector_search, py::object embed_one,
             float semantic_ratio) {
              HybridInput in;
              if (!query.is_none()) in.query = query.cast<std::string>();
              in.has_semantic = has_semantic;
              if (!vector.is_none()) in.vector = vector.cast<std::vector<float>>();
              in.offset = offset;
              in.limit = limit;
              in.keyword_search = [&]() { return result_from_py(keyword_search()); };
              in.vector_search = [&](const std::vector<float>& v) { return result_from_py(vector_search(v)); };
              if (!embed_one.is_none())
                  in.embed_one = [&](const std::string& q) -> std::optional<std::vector<float>> {
                      py::object r = embed_one(q);
                      if (r.is_none()) return std::nullopt;
                      return r.cast<std::vector<float>>();
                  };
              auto r = execute_hybrid(in, semantic_ratio);
              return py::make_tuple(result_to_py(r.first),
                                    r.second ? py::object(py::int_(*r.second)) : py::object(py::none()));
          });

    py::class_<GpuContext, std::shared_ptr<GpuContext>>(m, "GpuContext")
        .def(py::init([](uintptr_t ctx_handle, std::map<std::string, uint8_t> ids, py::object docids) {
            return std::make_shared<GpuContext>(reinterpret_cast<vs_ctx*>(ctx_handle), std::move(ids),
                                                to_docset(docids));
        }));

    m.def("assign_embedder_category_ids", &assign_embedder_category_ids);
    // One round of DocSet algebra for the tests: `a` and `b` are built as to_docset builds them (or
    // as the range [0, n) from an int), then `script` is played on a copy of `a`: ("insert", d),
    // ("remove", d), ("or", ), ("sub", ). Everything a DocSet can be asked is returned.
    m.def("docset_probe", [](const py::object& a_in, const py::object& b_in, const py::list& script,
                             const std::vector<DocumentId>& probes) {
        auto build = [](const py::object& o) {
            return py::isinstance<py::int_>(o) ? DocSet::from_range(o.cast<uint64_t>()) : to_docset(o);
        };
        const DocSet a = build(a_in), b = build(b_in);
        py::dict out;
        out["a_repr"] = a.representation();
        out["b_repr"] = b.representation();
        uint64_t n_and = 0;
        const DocSet both = a.intersect(b, &n_and);
        out["and"] = from_docset(both);
        out["and_len"] = n_and;
        out["and_repr"] = both.representation();
        out["and_len_again"] = both.len();
        out["or"] = from_docset(a | b);
        DocSet diff = a;
        diff -= b;
        out["sub"] = from_docset(diff);
        out["sub_len"] = diff.len();
        out["a_after_ops"] = from_docset(a);  // copies were modified, not `a`
        DocSet c = a;
        py::list effects;
        for (py::handle step : script) {
            const py::tuple t = py::reinterpret_borrow<py::tuple>(step);
            const std::string op = t[0].cast<std::string>();
            if (op == "insert") effects.append(c.insert(t[1].cast<DocumentId>()));
            else if (op == "remove") effects.append(c.remove(t[1].cast<DocumentId>()));
            else if (op == "or") c |= b;
            else if (op == "sub") c -= b;
            effects.append(c.len());
        }
        out["effects"] = effects;
        out["c"] = from_docset(c);
        out["c_repr"] = c.representation();
        out["c_empty"] = c.is_empty();
        py::list hit;
        for (DocumentId d : probes) hit.append(py::make_tuple(a.contains(d), c.contains(d)));
        out["contains"] = hit;
        const std::vector<uint64_t>& w = c.words();
        out["c_words"] = py::array_t<uint64_t>((py::ssize_t)w.size(), w.data());
        return out;
    });
    py::class_<GpuWriters>(m, "GpuWriters")
        .def(py::init([](uintptr_t ctx_handle, uint8_t embedder_index, uint32_t dimension, uint64_t capacity) {
            return new GpuWriters(reinterpret_cast<vs_ctx*>(ctx_handle), embedder_index, dimension, capacity);
        }));
    // writers: a GpuWriters, or a Python callable k -> fake writer object
    m.def("write_vector_points", [](py::object writers, const py::dict& points) {
        const VectorPoints p = points_from_py(points);
        if (py::isinstance<GpuWriters>(writers)) return write_vector_points(writers.cast<GpuWriters&>(), p);
        PyWriters w(writers);
        write_vector_points(w, p);
    });
    // Measurement hook: the write path for `n_docs` synthetic documents (one generated embedding each,
    // every 16th document two) as ONE chunk through write_vector_points and the build loop over
    // GpuWriters; returns the seconds the two calls took (generating the vectors is not timed).
    m.def("time_write_path", [](GpuWriters& writers, uint32_t dim, uint32_t n_docs) {
        VectorPoints p;
        p.embeddings.reserve(n_docs);
        uint32_t s = 0x9E3779B9u;
        for (uint32_t d = 0; d < n_docs; ++d) {
            std::vector<std::vector<float>> vecs(d % 16 == 0 ? 2 : 1, std::vector<float>(dim));
            for (auto& v : vecs)
                for (float& x : v) {
                    s ^= s << 13;
                    s ^= s >> 17;
                    s ^= s << 5;
                    x = (float)(int32_t)s * (1.0f / 2147483648.0f);
                }
            p.embeddings.emplace_back(d, std::move(vecs));
        }
        const auto t0 = std::chrono::steady_clock::now();
        write_vector_points(writers, p);
        build_vector_stores(writers);
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
    m.def("build_vector_stores", [](py::object writers) {
        if (py::isinstance<GpuWriters>(writers)) return build_vector_stores(writers.cast<GpuWriters&>());
        PyWriters w(writers);
        return build_vector_stores(w);
    });

    py::class_<VectorSortHolder>(m, "VectorSort")
        .def(py::init([](py::object ctx, std::vector<float> target, py::object vector_candidates,
                         size_t limit, const std::string& embedder_name, py::object distribution) {
            VectorSortHolder h;
            h.ctx = context_from_py(ctx);
            h.rule.reset(new VectorSort(*h.ctx, std::move(target), to_docset(vector_candidates), limit,
                                        embedder_name, distribution_from_py(distribution)));
            return h;
        }))
        .def("start_iteration",
             [](VectorSortHolder& h, py::object candidates) { h.rule->start_iteration(to_docset(candidates)); })
        .def("next_bucket", [](VectorSortHolder& h) { return bucket_to_py(h.rule->next_bucket()); });

    m.def("bucket_sort", [](const py::object& rules, uint64_t limit, uint64_t skip, py::object candidates) {
        std::vector<std::unique_ptr<RankingRule>> rr;
        for (py::handle h : rules) rr.push_back(rule_from_py(h));
        BucketSortOutput out = bucket_sort(rr, limit, skip, to_docset(candidates));
        SearchResult r;
        r.candidates = std::move(out.candidates);
        r.documents_ids = std::move(out.docids);
        r.document_scores = std::move(out.scores);
        return result_to_py(r);
    });

    m.def("execute_vector_search",
          [](py::object ctx, std::vector<float> vector, py::object universe, uint64_t skip, uint64_t limit,
             const std::string& embedder_name, py::object distribution, py::object extra_rules) {
              std::shared_ptr<Context> c = context_from_py(ctx);
              ExtraRules extra;
              if (!extra_rules.is_none())
                  extra = [&]() {
                      std::vector<std::unique_ptr<RankingRule>> v;
                      for (py::handle h : extra_rules()) v.push_back(rule_from_py(h));
                      return v;
                  };
              return result_to_py(execute_vector_search(*c, vector, to_docset(universe), skip, limit,
                                                        embedder_name, distribution_from_py(distribution),
                                                        extra));
          },
          py::arg("ctx"), py::arg("vector"), py::arg("universe"), py::arg("skip"), py::arg("limit"),
          py::arg("embedder_name"), py::arg("distribution") = py::none(),
          py::arg("extra_rules") = py::none());

    // one universe per vector (several requests in one call); universes[i] may be the same Python
    // object as universes[j]: those requests then share one C++ set, as a host's filter cache would
    m.def("execute_vector_search_batch_universes",
          [](py::object ctx, std::vector<std::vector<float>> vectors, py::list universes, uint64_t skip,
             uint64_t limit, const std::string& embedder_name, py::object distribution) {
              std::shared_ptr<Context> c = context_from_py(ctx);
              std::vector<DocSet> sets;
              std::vector<PyObject*> seen;
              for (py::handle u : universes) {
                  size_t at = seen.size();
                  for (size_t j = 0; j < seen.size(); ++j)
                      if (seen[j] == u.ptr()) at = j;
                  seen.push_back(u.ptr());
                  sets.push_back(at < sets.size() ? sets[at] : to_docset(py::reinterpret_borrow<py::object>(u)));
              }
              py::list out;
              for (const SearchResult& r : execute_vector_search_batch(*c, vectors, sets, skip, limit, embedder_name,
                                                                       distribution_from_py(distribution)))
                  out.append(result_to_py(r));
              return out;
          },
          py::arg("ctx"), py::arg("vectors"), py::arg("universes"), py::arg("skip"), py::arg("limit"),
          py::arg("embedder_name"), py::arg("distribution") = py::none());

    // execute_vector_search once per vector with ONE universe object, as a host does that keeps
    // the bitmap of a facet filter across requests (what a reader left with the set stays with it)
    m.def("execute_vector_search_same_universe",
          [](py::object ctx, std::vector<std::vector<float>> vectors, py::object universe, uint64_t skip,
             uint64_t limit, const std::string& embedder_name, py::object distribution) {
              std::shared_ptr<Context> c = context_from_py(ctx);
              const DocSet u = to_docset(universe);
              py::list out;
              for (const auto& v : vectors)
                  out.append(result_to_py(execute_vector_search(*c, v, u, skip, limit, embedder_name,
                                                                distribution_from_py(distribution))));
              return out;
          },
          py::arg("ctx"), py::arg("vectors"), py::arg("universe"), py::arg("skip"), py::arg("limit"),
          py::arg("embedder_name"), py::arg("distribution") = py::none());

    m.def("execute_vector_search_batch",
          [](py::object ctx, std::vector<std::vector<float>> vectors, py::object universe, uint64_t skip,
             uint64_t limit, const std::string& embedder_name, py::object distribution) {
              std::shared_ptr<Context> c = context_from_py(ctx);
              py::list out;
              for (const SearchResult& r : execute_vector_search_batch(*c, vectors, to_docset(universe), skip, limit,
                                                                       embedder_name, distribution_from_py(distribution)))
                  out.append(result_to_py(r));
              return out;
          },
          py::arg("ctx"), py::arg("vectors"), py::arg("universe"), py::arg("skip"), py::arg("limit"),
          py::arg("embedder_name"), py::arg("distribution") = py::none());
}
