// Host-side text I/O of the trial-list path (no device code): the callers either side of the scoring kernels.
//   * reading a trials / key file and resolving its id columns to x-vector table rows
//       utils/sv_trials_loaders.py:377-383, :400-406 (np.genfromtxt + a Python loop of dict look-ups per trial)
//       utils/sv_trials_loaders.py:429-437        (per-trial os.path.basename/splitext + dict look-up)
//   * writing the score files
//       utils/scorefile_generator.py:22-56         (astype(str) + np.c_ + np.savetxt, one Python-level row at a time)
// At 5e9 scored pairs/s the reference's text handling (~7e4 trials/s, measured) is the whole wall clock of score-file
// generation; these routines tokenise, hash and format at memory speed on one host thread and reproduce the
// reference's bytes: np.genfromtxt(dtype=str) tokenisation (any-whitespace delimiter, '#' comments, blank lines
// skipped, ragged rows rejected), os.path.basename / splitext id normalisation, str(np.float32) number formatting.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nplda_hip.h"

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

struct Tok { const char* p; size_t n; };

// Tokenise one line [p, e) (no '\n' inside); the comment tail is cut first.  Returns the number of tokens, the
// first `cap` of them in out.
inline int split_line(const char* p, const char* e, Tok* out, int cap) {
    const char* h = (const char*)memchr(p, '#', (size_t)(e - p));
    if (h) e = h;
    int n = 0;
    while (p < e) {
        while (p < e && is_space(*p)) ++p;
        if (p >= e) break;
        const char* q = p;
        while (q < e && !is_space(*q)) ++q;
        if (n < cap) out[n] = Tok{p, (size_t)(q - p)};
        ++n;
        p = q;
    }
    return n;
}

// os.path.basename: everything after the last '/'
inline Tok basename_of(Tok t) {
    for (size_t i = t.n; i > 0; --i)
        if (t.p[i - 1] == '/') return Tok{t.p + i, t.n - i};
    return t;
}

// os.path.splitext(...)[0]: cut at the last '.' of the last path component unless that component's leading
// characters up to it are all dots (".bashrc", "..x" keep their dots)
inline Tok strip_ext(Tok t) {
    size_t sep = 0;  // index just after the last '/'
    for (size_t i = t.n; i > 0; --i)
        if (t.p[i - 1] == '/') { sep = i; break; }
    size_t dot = (size_t)-1;
    for (size_t i = t.n; i > sep; --i)
        if (t.p[i - 1] == '.') { dot = i - 1; break; }
    if (dot == (size_t)-1) return t;
    size_t k = sep;
    while (k < dot && t.p[k] == '.') ++k;
    if (k == dot) return t;  // only dots before it: not an extension
    return Tok{t.p, dot};
}

// mode 3: str.replace('.sph', '') — every occurrence removed (utils/adaptive_score_normalization.py:25); needs a
// scratch buffer when something is removed
inline Tok normalise(Tok t, int mode, std::string* scratch = nullptr) {
    if (mode == 1) return strip_ext(t);
    if (mode == 2) return strip_ext(basename_of(t));
    if (mode == 3 && scratch && t.n >= 4) {
        const char* hit = nullptr;
        for (size_t i = 0; i + 4 <= t.n; ++i)
            if (memcmp(t.p + i, ".sph", 4) == 0) { hit = t.p + i; break; }
        if (!hit) return t;
        scratch->clear();
        for (size_t i = 0; i < t.n;) {
            if (i + 4 <= t.n && memcmp(t.p + i, ".sph", 4) == 0) i += 4;
            else scratch->push_back(t.p[i++]);
        }
        return Tok{scratch->data(), scratch->size()};
    }
    return t;
}

inline uint64_t hash_bytes(const char* p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a, then a finaliser (ids share long prefixes)
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h;
}

struct IdTable {
    const char* blob;
    const int64_t* off;
    std::vector<int64_t> slot;  // open addressing, -1 = empty, else id index
    uint64_t mask;
    IdTable(const char* b, const int64_t* o, int64_t n) : blob(b), off(o) {
        uint64_t cap = 16;
        while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
        mask = cap - 1;
        slot.assign(cap, -1);
        for (int64_t i = 0; i < n; ++i) {
            const char* p = blob + off[i];
            const size_t len = (size_t)(off[i + 1] - off[i] - 1);  // ids are '\n'-separated
            uint64_t h = hash_bytes(p, len) & mask;
            while (true) {
                const int64_t s = slot[h];
                if (s < 0) { slot[h] = i; break; }
                const size_t sl = (size_t)(off[s + 1] - off[s] - 1);
                if (sl == len && memcmp(blob + off[s], p, len) == 0) { slot[h] = i; break; }  // dict: last key wins
                h = (h + 1) & mask;
            }
        }
    }
    int64_t find(Tok t) const {
        uint64_t h = hash_bytes(t.p, t.n) & mask;
        while (true) {
            const int64_t s = slot[h];
            if (s < 0) return -1;
            const size_t sl = (size_t)(off[s + 1] - off[s] - 1);
            if (sl == t.n && memcmp(blob + off[s], t.p, t.n) == 0) return s;
            h = (h + 1) & mask;
        }
    }
};

// Python's float(str) on a token, the subset that appears in label columns: optional sign, decimal / exponent
// forms, inf / infinity / nan (case-insensitive).  Returns false when Python would raise ValueError.
inline bool parse_label(Tok t, float* out) {
    const char* p = t.p;
    const char* e = t.p + t.n;
    if (p == e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
    if (p == e || *p == '+' || *p == '-') return false;
    double v = 0.0;
    const auto r = std::from_chars(p, e, v, std::chars_format::general);
    if (r.ec == std::errc::result_out_of_range) {
        v = HUGE_VAL;  // float('1e999') == inf
        if (r.ptr != e) return false;
    } else if (r.ec != std::errc() || r.ptr != e) {
        return false;
    }
    *out = (float)(neg ? -v : v);
    return true;
}

// str(np.float32(v)) / str(np.float64(v)): shortest digits that round-trip in the type; positional for
// 1e-4 <= |v| < 1e16 (and 0) with at least one digit after the point, else scientific with a >= 2-digit exponent
// and no trailing ".0"
template <typename T>
int format_np(T v, char* out) {
    if (std::isnan(v)) { memcpy(out, "nan", 4); return 3; }
    if (std::isinf(v)) {
        if (v < 0) { memcpy(out, "-inf", 5); return 4; }
        memcpy(out, "inf", 4);
        return 3;
    }
    char* o = out;
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    if (v == (T)0) { memcpy(o, "0.0", 4); return (int)(o - out) + 3; }
    char sci[48];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);  // d[.ddd]e[+-]XX
    char digits[24];
    int nd = 0;
    const char* p = sci;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int ex = 0;
    {
        ++p;  // 'e'
        const bool eneg = *p == '-';
        ++p;
        for (; p < r.ptr; ++p) ex = ex * 10 + (*p - '0');
        if (eneg) ex = -ex;
    }
    if ((double)v >= 1e-4 && (double)v < 1e16) {  // numpy compares against the (long) double constants
        if (ex >= 0) {
            for (int i = 0; i <= ex; ++i) *o++ = i < nd ? digits[i] : '0';
            *o++ = '.';
            if (nd > ex + 1) for (int i = ex + 1; i < nd; ++i) *o++ = digits[i];
            else *o++ = '0';
        } else {
            *o++ = '0';
            *o++ = '.';
            for (int i = 0; i < -ex - 1; ++i) *o++ = '0';
            for (int i = 0; i < nd; ++i) *o++ = digits[i];
        }
    } else {
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            for (int i = 1; i < nd; ++i) *o++ = digits[i];
        }
        *o++ = 'e';
        *o++ = ex < 0 ? '-' : '+';
        const int ae = ex < 0 ? -ex : ex;
        if (ae < 10) *o++ = '0';
        o = std::to_chars(o, o + 4, ae).ptr;
    }
    *o = 0;
    return (int)(o - out);
}

// Python float(token): see parse_label; doubles for score columns ("astype(float)")
inline bool parse_f64(Tok t, double* out) {
    const char* p = t.p;
    const char* e = t.p + t.n;
    if (p == e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
    if (p == e || *p == '+' || *p == '-') return false;
    double v = 0.0;
    const auto r = std::from_chars(p, e, v, std::chars_format::general);
    if (r.ec == std::errc::result_out_of_range) {
        if (r.ptr != e) return false;
        // from_chars leaves v untouched: overflow -> inf, underflow -> 0 like strtod
        const std::string tmp(p, e);
        v = strtod(tmp.c_str(), nullptr);
    } else if (r.ec != std::errc() || r.ptr != e) {
        return false;
    }
    *out = neg ? -v : v;
    return true;
}

// visit the tokens of every data row after skip_rows; fn(row_index_after_skip, tokens, ntokens) -> false stops
template <typename F>
inline void for_rows(const char* text, size_t len, int64_t skip_rows, F&& fn) {
    const char* p = text;
    const char* end = text + len;
    int64_t row = 0;
    Tok tk[64];
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* e = nl ? nl : end;
        const int n = split_line(p, e, tk, 64);
        p = nl ? nl + 1 : end;
        if (n == 0) continue;
        const int64_t r = row++;
        if (r < skip_rows) continue;
        if (!fn(r - skip_rows, tk, n < 64 ? n : 64)) return;
    }
}

// ---- line-aligned chunks for the host threads ------------------------------------------------------------------------
// Large tables (a cfg3-scale cohort score file has 2.2e8 lines) are cut at line boundaries into one chunk per thread; a
// first parallel pass counts the data rows of every chunk, so that each thread knows the global index of its first row
// and all outputs keep the file order.  NPLDA_TEXT_THREADS overrides the thread count (default: min(16, cores), one
// thread per 2 MB of text at least).
struct Chunk {
    const char* b;
    const char* e;
    int64_t row0;   // global index (before skip_rows) of the chunk's first data row
    int64_t rows;   // data rows in the chunk
    int cols;       // columns of its rows, -1 if they disagree, 0 if it has none
};

inline int text_threads(size_t len) {
    int want = 0;
    if (const char* env = getenv("NPLDA_TEXT_THREADS")) want = atoi(env);
    if (want > 0) return want > 64 ? 64 : want;  // explicit: honoured as is (tests force several threads on tiny files)
    const unsigned hw = std::thread::hardware_concurrency();
    want = hw == 0 ? 4 : (hw > 16 ? 16 : (int)hw);
    const size_t by_size = len / (2u << 20) + 1;
    if ((size_t)want > by_size) want = (int)by_size;
    return want < 1 ? 1 : want;
}

template <typename F>
inline void run_threads(int n, F&& fn) {
    if (n <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)n - 1);
    for (int t = 1; t < n; ++t) th.emplace_back([&fn, t]() { fn(t); });
    fn(0);
    for (auto& x : th) x.join();
}

inline std::vector<Chunk> make_chunks(const char* text, size_t len) {
    const int nt = text_threads(len);
    std::vector<Chunk> ch((size_t)nt);
    const char* end = text + len;
    const char* p = text;
    for (int t = 0; t < nt; ++t) {
        const char* target = t + 1 == nt ? end : text + (len / (size_t)nt) * (size_t)(t + 1);
        const char* q = target;
        if (q < p) q = p;
        if (q < end) {
            const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q));
            q = nl ? nl + 1 : end;
        }
        ch[(size_t)t] = Chunk{p, q, 0, 0, 0};
        p = q;
    }
    run_threads(nt, [&](int t) {
        Chunk& c = ch[(size_t)t];
        const char* p2 = c.b;
        while (p2 < c.e) {
            const char* nl = (const char*)memchr(p2, '\n', (size_t)(c.e - p2));
            const char* e2 = nl ? nl : c.e;
            Tok dummy;
            const int n = split_line(p2, e2, &dummy, 1);
            if (n > 0) {
                if (c.rows == 0) c.cols = n;
                else if (n != c.cols) c.cols = -1;
                ++c.rows;
            }
            p2 = nl ? nl + 1 : c.e;
        }
    });
    int64_t r = 0;
    for (auto& c : ch) { c.row0 = r; r += c.rows; }
    return ch;
}

// visit the data rows of one chunk: fn(global_row_index_after_skip, tokens, ntokens) -> false stops this chunk
template <typename F>
inline void for_chunk_rows(const Chunk& c, int64_t skip_rows, F&& fn) {
    const char* p = c.b;
    int64_t row = c.row0;
    Tok tk[64];
    while (p < c.e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(c.e - p));
        const char* e = nl ? nl : c.e;
        const int n = split_line(p, e, tk, 64);
        p = nl ? nl + 1 : c.e;
        if (n == 0) continue;
        const int64_t r = row++;
        if (r < skip_rows) continue;
        if (!fn(r - skip_rows, tk, n < 64 ? n : 64)) return;
    }
}

}  // namespace

extern "C" {

int nplda_format_f32(float v, char* out) { return format_np<float>(v, out); }
int nplda_format_f64(double v, char* out) { return format_np<double>(v, out); }

int nplda_text_column_f64(const char* text, size_t len, int64_t skip_rows, int col, double* out, int64_t n) {
    if ((!text && len) || (!out && n) || n < 0 || col < -64 || col > 63) return NPLDA_EINVAL;
    const std::vector<Chunk> ch = make_chunks(text, len);
    const int64_t total = ch.empty() ? 0 : ch.back().row0 + ch.back().rows;
    if (total - (skip_rows < total ? skip_rows : total) != n) return NPLDA_EINVAL;  // fewer / more rows than asked
    std::vector<int> bad(ch.size(), 0);
    run_threads((int)ch.size(), [&](int t) {
        for_chunk_rows(ch[(size_t)t], skip_rows, [&](int64_t r, const Tok* tk, int nt) {
            const int c = col < 0 ? nt + col : col;
            if (c < 0 || c >= nt || !parse_f64(tk[c], &out[r])) { bad[(size_t)t] = 1; return false; }
            return true;
        });
    });
    for (int v : bad) if (v) return NPLDA_EINVAL;
    return NPLDA_OK;
}

namespace {
struct TokSet {
    std::vector<Tok> slot;
    uint64_t mask;
    int64_t count = 0;
    TokSet() : slot(1 << 12, Tok{nullptr, 0}), mask((1 << 12) - 1) {}
    static bool put(std::vector<Tok>& tab, uint64_t m, Tok t) {  // true if new
        uint64_t h = hash_bytes(t.p, t.n) & m;
        while (true) {
            Tok& s = tab[h];
            if (!s.p) { s = t; return true; }
            if (s.n == t.n && memcmp(s.p, t.p, t.n) == 0) return false;
            h = (h + 1) & m;
        }
    }
    void insert(Tok t) {
        if (!put(slot, mask, t)) return;
        if ((uint64_t)(++count) * 2 > mask) {
            std::vector<Tok> bigger(slot.size() * 2, Tok{nullptr, 0});
            const uint64_t bm = bigger.size() - 1;
            for (const Tok& x : slot)
                if (x.p) put(bigger, bm, x);
            slot.swap(bigger);
            mask = bm;
        }
    }
};
}  // namespace

int nplda_text_count_unique(const char* text, size_t len, int64_t skip_rows, int col, int64_t* n_unique) {
    if ((!text && len) || !n_unique || col < -64 || col > 63) return NPLDA_EINVAL;
    const std::vector<Chunk> ch = make_chunks(text, len);
    std::vector<TokSet> sets(ch.size());
    std::vector<int> bad(ch.size(), 0);
    run_threads((int)ch.size(), [&](int t) {
        TokSet& set = sets[(size_t)t];
        for_chunk_rows(ch[(size_t)t], skip_rows, [&](int64_t, const Tok* tk, int nt) {
            const int c = col < 0 ? nt + col : col;
            if (c < 0 || c >= nt) { bad[(size_t)t] = 1; return false; }
            set.insert(tk[c]);
            return true;
        });
    });
    for (int v : bad) if (v) return NPLDA_EINVAL;
    for (size_t t = 1; t < sets.size(); ++t)
        for (const Tok& x : sets[t].slot)
            if (x.p) sets[0].insert(x);
    *n_unique = sets.empty() ? 0 : sets[0].count;
    return NPLDA_OK;
}

int nplda_text_column_spans(const char* text, size_t len, int64_t skip_rows, int col, int64_t stride, int64_t* start,
                            int64_t* length, int64_t n) {
    if ((!text && len) || !start || !length || n < 0 || stride < 1 || col < -64 || col > 63) return NPLDA_EINVAL;
    int rc = NPLDA_OK;
    int64_t got = 0;
    for_rows(text, len, skip_rows, [&](int64_t r, const Tok* tk, int nt) {
        if (r % stride) return true;
        const int64_t k = r / stride;
        if (k >= n) return false;
        const int c = col < 0 ? nt + col : col;
        if (c < 0 || c >= nt) { rc = NPLDA_EINVAL; return false; }
        start[k] = tk[c].p - text;
        length[k] = (int64_t)tk[c].n;
        got = k + 1;
        return true;
    });
    if (rc == NPLDA_OK && got != n) rc = NPLDA_EINVAL;
    return rc;
}

int64_t nplda_text_scan(const char* text, size_t len, int* ncols) {
    if ((!text && len) || !ncols) return NPLDA_EINVAL;
    const std::vector<Chunk> ch = make_chunks(text, len);
    int cols = 0;
    int64_t rows = 0;
    for (const Chunk& c : ch) {
        if (c.rows == 0) continue;
        if (c.cols < 0 || (cols != 0 && c.cols != cols)) return NPLDA_EINVAL;  // genfromtxt: ragged rows are an error
        cols = c.cols;
        rows += c.rows;
    }
    *ncols = cols;
    return rows;
}

int nplda_text_lookup(const char* text, size_t len, int64_t skip_rows, int mode1, int mode2, int label_col,
                      const char* ids, const int64_t* id_off, const int64_t* id_num, int64_t n_ids, int64_t* i1,
                      int64_t* i2, float* label, int64_t* row_of, int64_t* n_kept, int64_t* first_bad_row) {
    if ((!text && len) || !ids || !id_off || n_ids < 0 || !i1 || !i2 || !n_kept) return NPLDA_EINVAL;
    if (mode1 < 0 || mode1 > 3 || mode2 < 0 || mode2 > 3 || label_col > 62) return NPLDA_EINVAL;
    if (label_col >= 0 && !label) return NPLDA_EINVAL;
    const IdTable tab(ids, id_off, n_ids);
    const std::vector<Chunk> ch = make_chunks(text, len);
    const int need = label_col >= 0 ? (label_col + 1 > 2 ? label_col + 1 : 2) : 2;
    struct Part { std::vector<int64_t> a, b, src; std::vector<float> lab; int64_t bad = -1; };
    std::vector<Part> parts(ch.size());
    run_threads((int)ch.size(), [&](int t) {
        Part& P = parts[(size_t)t];
        std::string scratch;
        for_chunk_rows(ch[(size_t)t], skip_rows, [&](int64_t r, const Tok* tk, int n) {
            bool ok = n >= need;
            int64_t a2 = -1, b2 = -1;
            float lab = 0.f;
            if (ok) {
                a2 = tab.find(normalise(tk[0], mode1, &scratch));
                b2 = tab.find(normalise(tk[1], mode2, &scratch));
                ok = a2 >= 0 && b2 >= 0;
                if (ok && label_col >= 0) ok = parse_label(tk[label_col], &lab);
            }
            if (!ok) {
                if (P.bad < 0) P.bad = r;
                return true;
            }
            P.a.push_back(id_num ? id_num[a2] : a2);
            P.b.push_back(id_num ? id_num[b2] : b2);
            if (label_col >= 0) P.lab.push_back(lab);
            P.src.push_back(r);
            return true;
        });
    });
    int64_t kept = 0, bad = -1;
    for (const Part& P : parts) {
        const size_t k = P.a.size();
        if (k) {
            memcpy(i1 + kept, P.a.data(), k * sizeof(int64_t));
            memcpy(i2 + kept, P.b.data(), k * sizeof(int64_t));
            if (label_col >= 0) memcpy(label + kept, P.lab.data(), k * sizeof(float));
            if (row_of) memcpy(row_of + kept, P.src.data(), k * sizeof(int64_t));
            kept += (int64_t)k;
        }
        if (bad < 0 && P.bad >= 0) bad = P.bad;
    }
    *n_kept = kept;
    if (first_bad_row) *first_bad_row = bad;
    return NPLDA_OK;
}

int nplda_scores_write(const char* path, const char* text, size_t len, int64_t skip_rows, int keep_cols,
                       const char* header, const void* scores, int scores_f64, int64_t n) {
    if (!path || (!text && len) || (!scores && n) || n < 0 || keep_cols < 1 || keep_cols > 64) return NPLDA_EINVAL;
    const std::vector<Chunk> ch = make_chunks(text, len);
    const int64_t total = ch.empty() ? 0 : ch.back().row0 + ch.back().rows;
    if (total - (skip_rows < total ? skip_rows : total) < n) return NPLDA_EINVAL;  // fewer data rows than scores
    std::vector<std::string> bufs(ch.size());
    std::vector<int> bad(ch.size(), 0);
    run_threads((int)ch.size(), [&](int t) {
        std::string& buf = bufs[(size_t)t];
        buf.reserve((size_t)(ch[(size_t)t].e - ch[(size_t)t].b) + (size_t)ch[(size_t)t].rows * 16);
        char num[48];
        for_chunk_rows(ch[(size_t)t], skip_rows, [&](int64_t r, const Tok* tk, int nt) {
            if (r >= n) return false;
            if (nt < keep_cols) { bad[(size_t)t] = 1; return false; }
            for (int c = 0; c < keep_cols; ++c) {
                buf.append(tk[c].p, tk[c].n);
                buf.push_back('\t');
            }
            const int k = scores_f64 ? format_np<double>(((const double*)scores)[r], num)
                                     : format_np<float>(((const float*)scores)[r], num);
            buf.append(num, (size_t)k);
            buf.push_back('\n');
            return true;
        });
    });
    for (int v : bad) if (v) return NPLDA_EINVAL;
    FILE* f = fopen(path, "wb");
    if (!f) return NPLDA_EINVAL;
    int rc = NPLDA_OK;
    if (header) {
        const size_t hl = strlen(header);
        if (fwrite(header, 1, hl, f) != hl || fputc('\n', f) == EOF) rc = NPLDA_EINVAL;
    }
    for (const std::string& buf : bufs)
        if (rc == NPLDA_OK && !buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) rc = NPLDA_EINVAL;
    if (fclose(f) != 0 && rc == NPLDA_OK) rc = NPLDA_EINVAL;
    return rc;
}

}  // extern "C"
