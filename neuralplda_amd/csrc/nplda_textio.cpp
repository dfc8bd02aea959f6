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
#include <string>
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

inline Tok normalise(Tok t, int mode) {
    if (mode == 1) return strip_ext(t);
    if (mode == 2) return strip_ext(basename_of(t));
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

}  // namespace

extern "C" {

int nplda_format_f32(float v, char* out) {
    // str(np.float32(v)): shortest digits that round-trip in float32; positional for 1e-4 <= |v| < 1e16 (and 0)
    // with at least one digit after the point, else scientific with a >= 2-digit exponent and no trailing ".0"
    if (std::isnan(v)) { memcpy(out, "nan", 4); return 3; }
    if (std::isinf(v)) {
        if (v < 0) { memcpy(out, "-inf", 5); return 4; }
        memcpy(out, "inf", 4);
        return 3;
    }
    char* o = out;
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    if (v == 0.0f) { memcpy(o, "0.0", 4); return (int)(o - out) + 3; }
    char sci[48];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);  // d[.ddd]e[+-]XX
    char digits[16];
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

int64_t nplda_text_scan(const char* text, size_t len, int* ncols) {
    if ((!text && len) || !ncols) return NPLDA_EINVAL;
    const char* p = text;
    const char* end = text + len;
    int64_t rows = 0;
    int cols = 0;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* e = nl ? nl : end;
        Tok dummy;
        const int n = split_line(p, e, &dummy, 1);
        if (n > 0) {
            if (rows == 0) cols = n;
            else if (n != cols) return NPLDA_EINVAL;  // genfromtxt: "Some errors were detected" (ragged rows)
            ++rows;
        }
        p = nl ? nl + 1 : end;
    }
    *ncols = cols;
    return rows;
}

int nplda_text_lookup(const char* text, size_t len, int64_t skip_rows, int mode1, int mode2, int label_col,
                      const char* ids, const int64_t* id_off, const int64_t* id_num, int64_t n_ids, int64_t* i1,
                      int64_t* i2, float* label, int64_t* row_of, int64_t* n_kept, int64_t* first_bad_row) {
    if ((!text && len) || !ids || !id_off || n_ids < 0 || !i1 || !i2 || !n_kept) return NPLDA_EINVAL;
    if (mode1 < 0 || mode1 > 2 || mode2 < 0 || mode2 > 2 || label_col > 62) return NPLDA_EINVAL;
    if (label_col >= 0 && !label) return NPLDA_EINVAL;
    const IdTable tab(ids, id_off, n_ids);
    const char* p = text;
    const char* end = text + len;
    int64_t row = 0, kept = 0, bad = -1;
    Tok tk[64];
    const int need = label_col >= 0 ? (label_col + 1 > 2 ? label_col + 1 : 2) : 2;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* e = nl ? nl : end;
        const int n = split_line(p, e, tk, 64);
        p = nl ? nl + 1 : end;
        if (n == 0) continue;
        const int64_t r = row++;
        if (r < skip_rows) continue;
        bool ok = n >= need;
        int64_t a = -1, b = -1;
        float lab = 0.f;
        if (ok) {
            a = tab.find(normalise(tk[0], mode1));
            b = tab.find(normalise(tk[1], mode2));
            ok = a >= 0 && b >= 0;
            if (ok && label_col >= 0) ok = parse_label(tk[label_col], &lab);
        }
        if (!ok) {
            if (bad < 0) bad = r - skip_rows;
            continue;
        }
        i1[kept] = id_num ? id_num[a] : a;
        i2[kept] = id_num ? id_num[b] : b;
        if (label_col >= 0) label[kept] = lab;
        if (row_of) row_of[kept] = r - skip_rows;
        ++kept;
    }
    *n_kept = kept;
    if (first_bad_row) *first_bad_row = bad;
    return NPLDA_OK;
}

int nplda_scores_write(const char* path, const char* text, size_t len, int64_t skip_rows, int keep_cols,
                       const char* header, const float* scores, int64_t n) {
    if (!path || (!text && len) || (!scores && n) || n < 0 || keep_cols < 1 || keep_cols > 64) return NPLDA_EINVAL;
    FILE* f = fopen(path, "wb");
    if (!f) return NPLDA_EINVAL;
    std::vector<char> buf;
    buf.reserve(1 << 22);
    auto flush = [&]() -> bool {
        const bool ok = buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        buf.clear();
        return ok;
    };
    if (header) {
        buf.insert(buf.end(), header, header + strlen(header));
        buf.push_back('\n');
    }
    const char* p = text;
    const char* end = text + len;
    int64_t row = 0, out = 0;
    Tok tk[64];
    char num[48];
    int rc = NPLDA_OK;
    while (p < end && out < n) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* e = nl ? nl : end;
        const int nt = split_line(p, e, tk, 64);
        p = nl ? nl + 1 : end;
        if (nt == 0) continue;
        if (row++ < skip_rows) continue;
        if (nt < keep_cols) { rc = NPLDA_EINVAL; break; }
        for (int c = 0; c < keep_cols; ++c) {
            buf.insert(buf.end(), tk[c].p, tk[c].p + tk[c].n);
            buf.push_back('\t');
        }
        const int k = nplda_format_f32(scores[out++], num);
        buf.insert(buf.end(), num, num + k);
        buf.push_back('\n');
        if (buf.size() > (1u << 22) - 4096 && !flush()) { rc = NPLDA_EINVAL; break; }
    }
    if (rc == NPLDA_OK && out != n) rc = NPLDA_EINVAL;  // fewer data rows than scores
    if (rc == NPLDA_OK && !flush()) rc = NPLDA_EINVAL;
    if (fclose(f) != 0 && rc == NPLDA_OK) rc = NPLDA_EINVAL;
    return rc;
}

}  // extern "C"
