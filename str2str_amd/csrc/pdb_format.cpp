// Native PDB text writer / merger for the exit of the sampling path (host code, no device work).
//
// At GPU sampling speed the Python f-string writer is the wall (BASELINE configs[3] writes
// 1024 x 512 x 5 = 2.6 M ATOM lines per t_delta).  These routines emit byte-for-byte the text of
//   protein.to_pdb              src/common/protein.py:152-234   (MODEL / ATOM / TER / ENDMDL, 80 columns, GLY has no CB)
//   pdb_utils.atom37_to_pdb     src/common/pdb_utils.py:205-252 (atom mask = sum |xyz| > 1e-7, one MODEL per replica,
//                                                                final "END" without a newline)
//   pdb_utils.merge_pdbfiles    src/common/pdb_utils.py:31-83   (MODELs renumbered in file order)
// and are pinned by the committed golden texts the reference's own writers produced (tests/test_io_cpu.py).
//
// Number formatting: Python's f"{x:>8.3f}" of a float32 coordinate is the correctly rounded (half-to-even) decimal of the
// binary value.  For a float32 x the product (double)x * 1000 is exact to far below the distance of any non-tie from a tie
// (x = n 2^-s, n < 2^24 => |x*1000 - (k + 1/2)| >= 2^-s whenever non-zero, while the double product is off by < 2^(-19-s)),
// and an exact tie is an exactly representable product, so nearbyint() in the default rounding mode reproduces it exactly.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "str2str_hip.h"

namespace {

const char* const kAtomTypes[37] = {"N",   "CA",  "C",   "CB",  "O",   "CG",  "CG1", "CG2", "OG",  "OG1", "SG",  "CD",  "CD1",
                                    "CD2", "ND1", "ND2", "OD1", "OD2", "SD",  "CE",  "CE1", "CE2", "CE3", "NE",  "NE1", "NE2",
                                    "OE1", "OE2", "CH2", "NH1", "NH2", "OH",  "CZ",  "CZ2", "CZ3", "NZ",  "OXT"};
const char* const kRes3[21] = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU",
                               "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL", "UNK"};
const char kChainIds[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789";

struct Sink {  // counts when buf == nullptr
    char* buf;
    long long cap, n;
    void put(const char* s, long long len) {
        if (buf && n + len <= cap) memcpy(buf + n, s, (size_t)len);
        n += len;
    }
};

// right-aligned decimal integer, width w (wider numbers expand, as Python's :>w does); returns chars written
inline int put_int(char* d, long long v, int w) {
    char tmp[24];
    int k = 0;
    const bool neg = v < 0;
    unsigned long long u = neg ? (unsigned long long)(-v) : (unsigned long long)v;
    do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) tmp[k++] = '-';
    int o = 0;
    for (int p = k; p < w; ++p) d[o++] = ' ';
    while (k) d[o++] = tmp[--k];
    return o;
}

// f"{x:>{w}.{dec}f}" for a float32 x (dec = 2 or 3); returns chars written
inline int put_fixed(char* d, float x, int w, int dec) {
    if (!std::isfinite(x)) {  // Python prints 'nan' / 'inf' right-aligned
        const char* s = std::isnan(x) ? "nan" : (x < 0 ? "-inf" : "inf");
        const int len = (int)strlen(s);
        int o = 0;
        for (int p = len; p < w; ++p) d[o++] = ' ';
        memcpy(d + o, s, (size_t)len);
        return o + len;
    }
    const double scale = dec == 3 ? 1000.0 : 100.0;
    const double y = std::nearbyint(std::fabs((double)x) * scale);  // half-to-even, exact (see the header comment)
    unsigned long long u = (unsigned long long)y;
    char tmp[32];
    int k = 0;
    for (int i = 0; i < dec; ++i) { tmp[k++] = (char)('0' + u % 10); u /= 10; }
    tmp[k++] = '.';
    do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (std::signbit(x)) tmp[k++] = '-';  // "-0.000" for negative values that round to zero, like Python
    int o = 0;
    for (int p = k; p < w; ++p) d[o++] = ' ';
    while (k) d[o++] = tmp[--k];
    return o;
}

inline void put_line(Sink& s, char* line, int len) {  // line.ljust(80) + "\n"
    while (len < 80) line[len++] = ' ';
    line[len++] = '\n';
    s.put(line, len);
}

inline int ter_line(char* L, long long atom_index, const char* res3, char chain, long long resi) {
    // f"{'TER':<6}{atom_index:>5}      {end_resname:>3} {chain_name:>1}{residue_index:>4}"
    int o = 0;
    memcpy(L, "TER   ", 6); o = 6;
    o += put_int(L + o, atom_index, 5);
    memcpy(L + o, "      ", 6); o += 6;
    memcpy(L + o, res3, 3); o += 3;
    L[o++] = ' ';
    L[o++] = chain;
    o += put_int(L + o, resi, 4);
    return o;
}

int format_models(Sink& s, const float* atom37, int n_models, int n_res, const long long* aatype, const long long* residue_index,
                  const long long* chain_index, const double* b_factors, int first_model, int add_end) {
    char L[256];
    for (int i = 0; i < n_res; ++i) {
        if (aatype && (aatype[i] < 0 || aatype[i] > 20)) return -1;          // "Invalid aatypes."
        if (chain_index && (chain_index[i] < 0 || chain_index[i] >= 62)) return -2;  // more than PDB_MAX_CHAINS chains
    }
    for (int m = 0; m < n_models; ++m) {
        const float* P = atom37 + (long long)m * n_res * 37 * 3;
        int o = 0;
        memcpy(L, "MODEL     ", 10); o = 10;
        o += put_int(L + o, (long long)first_model + m, 1);
        put_line(s, L, o);
        long long atom_index = 1;
        long long last_chain = chain_index ? chain_index[0] : 0;
        for (int i = 0; i < n_res; ++i) {
            const long long ch = chain_index ? chain_index[i] : 0;
            const long long aa = aatype ? aatype[i] : 0;
            const long long resi = residue_index ? residue_index[i] : (long long)i + 1;
            if (ch != last_chain) {
                const long long pa = aatype ? aatype[i - 1] : 0, pr = residue_index ? residue_index[i - 1] : (long long)i;
                put_line(s, L, ter_line(L, atom_index, kRes3[pa], kChainIds[chain_index[i - 1]], pr));
                last_chain = ch;
                ++atom_index;
            }
            const char* res3 = kRes3[aa];
            const bool gly = aa == 7;
            for (int a = 0; a < 37; ++a) {
                const float x = P[((long long)i * 37 + a) * 3 + 0], y = P[((long long)i * 37 + a) * 3 + 1],
                            z = P[((long long)i * 37 + a) * 3 + 2];
                // mask = np.sum(np.abs(pos37), axis=-1) > 1e-7 in float32 (pdb_utils.py:233); NaN compares false
                const float ssum = (std::fabs(x) + std::fabs(y)) + std::fabs(z);
                if (!(ssum > 1e-7f) || (gly && a == 3)) continue;
                const char* an = kAtomTypes[a];
                const int al = (int)strlen(an);
                o = 0;
                memcpy(L, "ATOM  ", 6); o = 6;
                o += put_int(L + o, atom_index, 5);
                L[o++] = ' ';
                // name = atom_name if len(atom_name) == 4 else f" {atom_name}", then {name:<4}
                int w = 0;
                if (al < 4) { L[o++] = ' '; w = 1; }
                memcpy(L + o, an, (size_t)al); o += al; w += al;
                for (; w < 4; ++w) L[o++] = ' ';
                L[o++] = ' ';                       // {'':>1}
                memcpy(L + o, res3, 3); o += 3;     // {name3:>3}
                L[o++] = ' ';
                L[o++] = kChainIds[ch];
                o += put_int(L + o, resi, 4);
                L[o++] = ' ';                       // {'':>1}
                L[o++] = ' '; L[o++] = ' '; L[o++] = ' ';
                o += put_fixed(L + o, x, 8, 3);
                o += put_fixed(L + o, y, 8, 3);
                o += put_fixed(L + o, z, 8, 3);
                memcpy(L + o, "  1.00", 6); o += 6;  // {1.0:>6.2f}
                const double bf = b_factors ? b_factors[(long long)i * 37 + a] : 0.0;
                if (bf == 0.0 && !std::signbit(bf)) { memcpy(L + o, "  0.00", 6); o += 6; }
                else o += snprintf(L + o, 40, "%6.2f", bf);
                memcpy(L + o, "          ", 10); o += 10;
                L[o++] = ' '; L[o++] = an[0];        // {atom_name[0]:>2}
                L[o++] = ' '; L[o++] = ' ';          // {'':>2}
                put_line(s, L, o);
                ++atom_index;
            }
        }
        {
            const long long pa = aatype ? aatype[n_res - 1] : 0, pr = residue_index ? residue_index[n_res - 1] : (long long)n_res;
            const long long pc = chain_index ? chain_index[n_res - 1] : 0;
            put_line(s, L, ter_line(L, atom_index, kRes3[pa], kChainIds[pc], pr));
        }
        memcpy(L, "ENDMDL", 6);
        put_line(s, L, 6);
        if (add_end == 1) { memcpy(L, "END", 3); put_line(s, L, 3); }
    }
    if (add_end == 2) s.put("END", 3);  // atom37_to_pdb: one bare END after the last model, no newline
    return 0;
}

}  // namespace

extern "C" long long s2s_format_pdb_models(const float* atom37, int n_models, int n_res, const long long* aatype,
                                           const long long* residue_index, const long long* chain_index,
                                           const double* b_factors, int first_model_number, int add_end, char* out,
                                           long long out_capacity) {
    if (!atom37 || n_models < 0 || n_res <= 0) return -1;
    Sink s{out, out ? out_capacity : 0, 0};
    const int rc = format_models(s, atom37, n_models, n_res, aatype, residue_index, chain_index, b_factors, first_model_number, add_end);
    return rc < 0 ? rc : s.n;
}

extern "C" long long s2s_write_pdb_models(const char* path, int append, const float* atom37, int n_models, int n_res,
                                          const long long* aatype, const long long* residue_index,
                                          const long long* chain_index, const double* b_factors, int first_model_number,
                                          int add_end) {
    if (!path || !atom37 || n_models < 0 || n_res <= 0) return -1;
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) return -3;
    // stream in blocks of models: bounded memory however many replicas are written
    const long long per_model = ((long long)n_res * 37 + 4) * 81 + 8;
    const int block = (int)std::max<long long>(1, std::min<long long>(n_models, (64LL << 20) / per_model));
    std::vector<char> buf((size_t)(per_model * block + 8));
    long long total = 0;
    for (int m0 = 0; m0 < n_models || (m0 == 0 && n_models == 0); m0 += block) {
        const int nm = std::min(block, n_models - m0);
        const bool last = m0 + nm >= n_models;
        Sink s{buf.data(), (long long)buf.size(), 0};
        const int rc = format_models(s, atom37 + (long long)m0 * n_res * 37 * 3, nm, n_res, aatype, residue_index, chain_index,
                                     b_factors, first_model_number + m0, add_end == 2 ? (last ? 2 : 0) : add_end);
        if (rc < 0 || s.n > (long long)buf.size() || fwrite(buf.data(), 1, (size_t)s.n, f) != (size_t)s.n) {
            fclose(f);
            return rc < 0 ? rc : -4;
        }
        total += s.n;
        if (n_models == 0) break;
    }
    if (fclose(f) != 0) return -4;
    return total;
}

extern "C" long long s2s_merge_pdb_files(const char* const* paths, int n_paths, const char* out_path) {
    // pdb_utils.merge_pdbfiles (:31-83): concatenate MODELs of every input in order, renumbered from 1; inputs without
    // MODEL records count as one model each; every kept line is stripped and left-justified to 80 columns.
    if (!paths || n_paths < 0 || !out_path) return -1;
    FILE* fo = fopen(out_path, "wb");
    if (!fo) return -3;
    std::vector<char> obuf;
    obuf.reserve(8u << 20);
    long long total = 0, model_number = 0;
    auto flush = [&]() {
        if (!obuf.empty()) { total += (long long)fwrite(obuf.data(), 1, obuf.size(), fo); obuf.clear(); }
    };
    auto emit = [&](const char* s, size_t len) {  // x.ljust(80) + "\n"
        obuf.insert(obuf.end(), s, s + len);
        for (size_t p = len; p < 80; ++p) obuf.push_back(' ');
        obuf.push_back('\n');
        if (obuf.size() > (6u << 20)) flush();
    };
    auto emit_model = [&]() {
        char L[40];
        int o = 10;
        memcpy(L, "MODEL     ", 10);
        o += put_int(L + o, model_number, 1);
        emit(L, (size_t)o);
    };
    std::string data;
    for (int p = 0; p < n_paths; ++p) {
        FILE* fi = fopen(paths[p], "rb");
        if (!fi) { fclose(fo); return -3; }
        data.clear();
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), fi)) > 0) data.append(chunk, got);
        fclose(fi);
        // first pass: does the file carry MODEL / ENDMDL records?
        bool has_models = false;
        for (size_t i = 0; i < data.size();) {
            size_t e = data.find('\n', i);
            if (e == std::string::npos) e = data.size();
            if (data.compare(i, 5, "MODEL") == 0 || data.compare(i, 6, "ENDMDL") == 0) { has_models = true; break; }
            i = e + 1;
        }
        if (!has_models) {
            ++model_number;
            emit_model();
        }
        for (size_t i = 0; i < data.size();) {
            size_t e = data.find('\n', i);
            if (e == std::string::npos) e = data.size();
            const char* ln = data.data() + i;
            const size_t len = e - i;
            const bool is_atom = len >= 4 && memcmp(ln, "ATOM", 4) == 0, is_ter = len >= 3 && memcmp(ln, "TER", 3) == 0;
            if (has_models && len >= 5 && memcmp(ln, "MODEL", 5) == 0) {
                ++model_number;
                if (model_number > 1) emit("ENDMDL", 6);
                emit_model();
            } else if (has_models && len >= 3 && memcmp(ln, "END", 3) == 0) {
                // END / ENDMDL of the input are dropped
            } else if (is_atom || is_ter) {
                size_t a = 0, b = len;  // str.strip()
                while (a < b && (ln[a] == ' ' || ln[a] == '\t' || ln[a] == '\r' || ln[a] == '\f' || ln[a] == '\v')) ++a;
                while (b > a && (ln[b - 1] == ' ' || ln[b - 1] == '\t' || ln[b - 1] == '\r' || ln[b - 1] == '\f' || ln[b - 1] == '\v')) --b;
                emit(ln + a, b - a);
            }
            i = e + 1;
        }
        if (!has_models) emit("ENDMDL", 6);
    }
    emit("ENDMDL", 6);
    emit("END", 3);
    flush();
    if (fclose(fo) != 0) return -4;
    return total;
}
