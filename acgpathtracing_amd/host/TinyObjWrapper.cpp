// TinyObjWrapper.cpp — see TinyObjWrapper.h.  The parser restates the behaviour of
// tinyobjloader v2.0.0 for the subset the reference consumes (positions, faces,
// usemtl/mtllib, Kd/Ke/Ni/Pr/Pm), including its decimal-to-double conversion
// (util/tiny_obj_loader.h:897-1023) so vertex coordinates come out bit-identical.
#include "TinyObjWrapper.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <thread>

namespace acgpt {
namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return (unsigned)(c - '0') < 10u; }
inline bool is_eol(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// Decimal text -> double exactly the way tinyobjloader does it: digits accumulated in a
// double, fraction digits added as digit * 10^-k (table for k < 8, pow() beyond), exponent
// applied as ldexp(m * 5^e, e).  Not correctly rounded — but it is what the reference
// feeds to the GPU, so we reproduce it instead of calling strtod.
bool parse_double(const char* s, const char* end, double* out)
{
    if (s >= end) return false;
    double mant = 0.0;
    int expo = 0;
    char sign = '+', esign = '+';
    const char* c = s;
    int read = 0;
    bool lead_dot = false;
    if (*c == '+' || *c == '-') {
        sign = *c++;
        if (c != end && *c == '.') lead_dot = true;
    } else if (is_digit(*c)) {
    } else if (*c == '.') {
        lead_dot = true;
    } else {
        return false;
    }
    bool more = (c != end);
    if (!lead_dot) {
        while (more && is_digit(*c)) { mant *= 10; mant += (int)(*c - '0'); c++; read++; more = (c != end); }
        if (read == 0) return false;
    }
    if (more) {
        bool go_exp = false;
        if (*c == '.') {
            c++; read = 1; more = (c != end);
            static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            while (more && is_digit(*c)) {
                mant += (int)(*c - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
                read++; c++; more = (c != end);
            }
            go_exp = more;
        } else if (*c == 'e' || *c == 'E') {
            go_exp = true;
        }
        if (go_exp && (*c == 'e' || *c == 'E')) {
            c++; more = (c != end);
            if (more && (*c == '+' || *c == '-')) { esign = *c++; }
            else if (is_digit(*c)) {}
            else return false;
            read = 0; more = (c != end);
            while (more && is_digit(*c)) {
                if (expo > 2147483647 / 10) return false;
                expo = expo * 10 + (int)(*c - '0');
                c++; read++; more = (c != end);
            }
            expo *= (esign == '+' ? 1 : -1);
            if (read == 0) return false;
        }
    }
    *out = (sign == '+' ? 1 : -1) * (expo ? std::ldexp(mant * std::pow(5.0, expo), expo) : mant);
    return true;
}

// next whitespace-delimited real, or `dflt` when absent / malformed
float next_real(const char** tok, double dflt = 0.0)
{
    *tok += strspn(*tok, " \t");
    const char* end = *tok + strcspn(*tok, " \t\r");
    double v = dflt;
    parse_double(*tok, end, &v);
    *tok = end;
    return (float)v;
}

std::string next_word(const char** tok)
{
    *tok += strspn(*tok, " \t");
    size_t n = strcspn(*tok, " \t\r");
    std::string s(*tok, *tok + n);
    *tok += n;
    return s;
}

// OBJ index -> zero-based (1-based positive, negative = relative to the current count)
bool fix_index(int idx, int n, int* out)
{
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx == 0) return false;
    *out = n + idx;
    return *out >= 0;
}

bool get_line(std::istream& is, std::string& line)
{
    line.clear();
    if (is.peek() == EOF) return false;
    std::getline(is, line);
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    return true;
}

struct RawMaterial {
    std::string name;
    float diffuse[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float ior = 1.0f, roughness = 0.0f, metallic = 0.0f;
};

// MTL subset: newmtl, Kd, Ke, Ni, Pr, Pm, map_Kd (only for tinyobj's "Kd defaults to 0.6
// when only map_Kd is given").  As in tinyobj the trailing material is always flushed, so
// any readable .mtl yields at least one record.
void load_mtl(std::istream& is, std::vector<RawMaterial>& mats, std::map<std::string, int>& by_name)
{
    RawMaterial cur;
    bool has_kd = false;
    std::string line;
    while (get_line(is, line)) {
        size_t last = line.find_last_not_of(" \t");
        line = (last == std::string::npos) ? std::string() : line.substr(0, last + 1);
        if (line.empty()) continue;
        const char* t = line.c_str();
        t += strspn(t, " \t");
        if (t[0] == '\0' || t[0] == '#') continue;
        if (0 == strncmp(t, "newmtl", 6) && is_space(t[6])) {
            if (!cur.name.empty()) {
                by_name.insert(std::make_pair(cur.name, (int)mats.size()));
                mats.push_back(cur);
            }
            cur = RawMaterial();
            has_kd = false;
            cur.name = std::string(t + 7);
            continue;
        }
        if (t[0] == 'K' && t[1] == 'd' && is_space(t[2])) {
            t += 2; cur.diffuse[0] = next_real(&t); cur.diffuse[1] = next_real(&t); cur.diffuse[2] = next_real(&t);
            has_kd = true; continue;
        }
        if (t[0] == 'K' && t[1] == 'e' && is_space(t[2])) {
            t += 2; cur.emission[0] = next_real(&t); cur.emission[1] = next_real(&t); cur.emission[2] = next_real(&t);
            continue;
        }
        if (t[0] == 'N' && t[1] == 'i' && is_space(t[2])) { t += 2; cur.ior = next_real(&t); continue; }
        if (t[0] == 'P' && t[1] == 'r' && is_space(t[2])) { t += 2; cur.roughness = next_real(&t); continue; }
        if (t[0] == 'P' && t[1] == 'm' && is_space(t[2])) { t += 2; cur.metallic = next_real(&t); continue; }
        if (0 == strncmp(t, "map_Kd", 6) && is_space(t[6])) {
            if (!has_kd) { cur.diffuse[0] = cur.diffuse[1] = cur.diffuse[2] = 0.6f; }
            continue;
        }
    }
    by_name.insert(std::make_pair(cur.name, (int)mats.size()));
    mats.push_back(cur);
}

// point-in-triangle by crossing number on the two projected axes
int pnpoly3(const float* vx, const float* vy, float tx, float ty)
{
    int c = 0;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
    return c;
}

struct Mesh { std::vector<uint32_t> idx; std::vector<uint32_t> mat; };

inline void emit_tri(Mesh& m, int a, int b, int c, int mat)
{
    m.idx.push_back((uint32_t)a); m.idx.push_back((uint32_t)b); m.idx.push_back((uint32_t)c);
    m.mat.push_back((uint32_t)mat);
}

// Triangulate faces [f0, f1) (corner lists corner[first[f]] .. corner[first[f + 1]]) against the vertex
// array as it stood when the flush was reached: `n_v` vertices, xyzw records (tinyobj triangulates
// when a group is flushed, not when the `f` line is read, so a face may use vertices that follow it).
void flush_faces(const int* corner, const uint32_t* first, size_t f0, size_t f1, int mat,
                 const float* v, size_t n_v, Mesh& out, std::string& warn)
{
    for (size_t fi = f0; fi < f1; fi++) {
        const int* fv = corner + first[fi];
        const size_t n = first[fi + 1] - first[fi];
        if (n < 3) { warn += "Degenerated face found\n."; continue; }
        if (n == 3) { emit_tri(out, fv[0], fv[1], fv[2], mat); continue; }
        if (n == 4) {
            size_t a = (size_t)fv[0], b = (size_t)fv[1], c = (size_t)fv[2], d = (size_t)fv[3];
            if (a >= n_v || b >= n_v || c >= n_v || d >= n_v) {
                warn += "Face with invalid vertex index found.\n";
                continue;
            }
            // split along the shorter diagonal: |v2-v0|^2 < |v3-v1|^2 -> (0,1,2)(0,2,3), else (0,1,3)(1,2,3)
            float acx = v[4 * c] - v[4 * a], acy = v[4 * c + 1] - v[4 * a + 1], acz = v[4 * c + 2] - v[4 * a + 2];
            float bdx = v[4 * d] - v[4 * b], bdy = v[4 * d + 1] - v[4 * b + 1], bdz = v[4 * d + 2] - v[4 * b + 2];
            float d02 = acx * acx + acy * acy + acz * acz;
            float d13 = bdx * bdx + bdy * bdy + bdz * bdz;
            if (d02 < d13) { emit_tri(out, fv[0], fv[1], fv[2], mat); emit_tri(out, fv[0], fv[2], fv[3], mat); }
            else           { emit_tri(out, fv[0], fv[1], fv[3], mat); emit_tri(out, fv[1], fv[2], fv[3], mat); }
            continue;
        }
        // n > 4: ear clipping in the plane of the first non-degenerate corner
        size_t ax0 = 1, ax1 = 2;
        for (size_t k = 0; k < n; k++) {
            size_t i0 = (size_t)fv[k % n], i1 = (size_t)fv[(k + 1) % n], i2 = (size_t)fv[(k + 2) % n];
            if (i0 >= n_v || i1 >= n_v || i2 >= n_v) continue;
            float e0x = v[4 * i1] - v[4 * i0], e0y = v[4 * i1 + 1] - v[4 * i0 + 1], e0z = v[4 * i1 + 2] - v[4 * i0 + 2];
            float e1x = v[4 * i2] - v[4 * i1], e1y = v[4 * i2 + 1] - v[4 * i1 + 1], e1z = v[4 * i2 + 2] - v[4 * i1 + 2];
            float cx = std::fabs(e0y * e1z - e0z * e1y);
            float cy = std::fabs(e0z * e1x - e0x * e1z);
            float cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = std::numeric_limits<float>::epsilon();
            if (cx > eps || cy > eps || cz > eps) {
                if (!(cx > cy && cx > cz)) { ax0 = 0; if (cz > cx && cz > cy) ax1 = 1; }
                break;
            }
        }
        std::vector<int> rem(fv, fv + n);
        size_t guess = 0;
        size_t budget = n;
        size_t prev_n = rem.size();
        while (rem.size() > 3 && budget > 0) {
            size_t m = rem.size();
            if (guess >= m) guess -= m;
            if (prev_n != m) { prev_n = m; budget = m; } else { budget--; }
            int ind[3]; float vx[3], vy[3];
            for (size_t k = 0; k < 3; k++) {
                ind[k] = rem[(guess + k) % m];
                size_t vi = (size_t)ind[k];
                if (vi >= n_v) { vx[k] = 0.0f; vy[k] = 0.0f; }
                else { vx[k] = v[vi * 4 + ax0]; vy[k] = v[vi * 4 + ax1]; }
            }
            float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
            float crs = e0x * e1y - e0y * e1x;
            float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
            if (crs * area < 0.0f) { guess += 1; continue; }        // reflex corner
            bool overlap = false;
            for (size_t o = 3; o < m; ++o) {
                size_t id = (guess + o) % m;
                if (id >= rem.size()) continue;
                size_t ovi = (size_t)rem[id];
                if (ovi >= n_v) continue;
                if (pnpoly3(vx, vy, v[ovi * 4 + ax0], v[ovi * 4 + ax1])) { overlap = true; break; }
            }
            if (overlap) { guess += 1; continue; }
            emit_tri(out, ind[0], ind[1], ind[2], mat);             // an ear: cut its middle vertex
            size_t gone = (guess + 1) % m;
            rem.erase(rem.begin() + (long)gone);
        }
        if (rem.size() == 3) emit_tri(out, rem[0], rem[1], rem[2], mat);
    }
}

// ---- three-pass ingest ---------------------------------------------------------------------------
// The file is read whole and cut at line ends into chunks.
//   pass A (parallel, read-only): per chunk, count lines, v / vn / vt statements, faces and face corners.
//           Prefix sums then give every chunk its place in the output arrays, which are allocated once
//           (first-touch page faults, not parsing, dominate a naive loader on a VM).
//   pass B (parallel): tokenise.  Vertices go straight into the final xyzw array, face corners (raw OBJ
//           indices) into one flat array; usemtl / mtllib / g / o become events, and a marker records the
//           v / vn / vt counts whenever they changed before a face.
//   pass C (one thread, cheap): replay events and faces in file order with the running counts — relative
//           indices, material look-up, tinyobj's flush points for triangulation, warnings and the first
//           error all depend on order and happen here.
// The result is the same, byte for byte, as reading the file line by line.
enum EventKind { EV_USEMTL, EV_MTLLIB, EV_GROUP };

struct Event {
    EventKind kind;
    size_t face_before;             // faces of this chunk that precede the statement
    uint32_t v_before, vn_before, vt_before;   // counts inside the chunk before this line
    size_t text;                    // index into Chunk::text
};

struct Marker { size_t face; uint32_t v, vn, vt; };   // from local face `face` on, the chunk-local counts are these

struct Chunk {
    size_t begin = 0, end = 0;
    // pass A
    uint32_t n_v = 0, n_vn = 0, n_vt = 0, n_lines = 0;
    size_t n_faces = 0, n_corners = 0;
    // prefix sums
    size_t base_v = 0, base_vn = 0, base_vt = 0, base_line = 0, base_face = 0, base_corner = 0;
    // pass B
    std::vector<Event> events;
    std::vector<Marker> marks;
    std::vector<std::string> text;
    std::vector<int> vt, vn;        // raw vt / vn indices per local corner; allocated when the first one appears
    bool overflow = false;          // pass B found more than pass A counted (internal error)
};

// corners of an `f` statement, t just behind "f" + blank; e = end of the line.  Same stepping as the
// tokeniser below, on a line that is not NUL-terminated yet.
inline bool in_set(char c, const char* set) { for (; *set; ++set) if (c == *set) return true; return false; }
inline const char* skip_set(const char* t, const char* e, const char* set) { while (t < e && *t && in_set(*t, set)) ++t; return t; }
inline const char* skip_until(const char* t, const char* e, const char* set) { while (t < e && *t && !in_set(*t, set)) ++t; return t; }
inline char at(const char* t, const char* e) { return t < e ? *t : '\0'; }

size_t count_corners(const char* t, const char* e)
{
    size_t n = 0;
    t = skip_set(t, e, " \t");
    while (!is_eol(at(t, e))) {
        t = skip_until(t, e, "/ \t\r");
        if (at(t, e) == '/') {
            t++;
            if (at(t, e) == '/') { t++; t = skip_until(t, e, "/ \t\r"); }
            else {
                t = skip_until(t, e, "/ \t\r");
                if (at(t, e) == '/') { t++; t = skip_until(t, e, "/ \t\r"); }
            }
        }
        n++;
        t = skip_set(t, e, " \t\r");
    }
    return n;
}

void count_chunk(const char* buf, Chunk& c)
{
    const char* p = buf + c.begin;
    const char* end = buf + c.end;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* e = nl ? nl : end;
        c.n_lines++;
        const char* t = skip_set(p, e, " \t");
        const char t0 = at(t, e), t1 = at(t + 1, e), t2 = at(t + 2, e);
        if (t0 == 'v' && is_space(t1)) c.n_v++;
        else if (t0 == 'v' && t1 == 'n' && is_space(t2)) c.n_vn++;
        else if (t0 == 'v' && t1 == 't' && is_space(t2)) c.n_vt++;
        else if (t0 == 'f' && is_space(t1)) { c.n_faces++; c.n_corners += count_corners(t + 2, e); }
        p = e + 1;
    }
}

struct Outputs {
    float* vertices;                // xyzw per vertex
    int* corner;                    // raw OBJ vertex index per face corner
    uint32_t* face_first;           // first corner of each face (+ one past the end)
    uint32_t* face_line;            // 1-based line number of each face inside its chunk
};

// One line, NUL-terminated in place.  Mirrors the dispatch order of a line-by-line loader.
struct ChunkCursor { uint32_t v = 0, vn = 0, vt = 0, line = 0; size_t face = 0, corner = 0; uint32_t mv = 0xFFFFFFFFu, mvn = 0, mvt = 0; };

void tokenise_line(const char* t, Chunk& c, ChunkCursor& k, const Outputs& o)
{
    t += strspn(t, " \t");
    if (t[0] == '\0' || t[0] == '#') return;
    if (t[0] == 'v' && is_space(t[1])) {
        t += 2;
        float x = next_real(&t), y = next_real(&t), z = next_real(&t);
        if (k.v >= c.n_v) { c.overflow = true; return; }
        float* dst = o.vertices + 4 * (c.base_v + k.v);
        dst[0] = x; dst[1] = y; dst[2] = z; dst[3] = 1.0f;
        k.v++;
        return;
    }
    if (t[0] == 'v' && t[1] == 'n' && is_space(t[2])) { k.vn++; return; }
    if (t[0] == 'v' && t[1] == 't' && is_space(t[2])) { k.vt++; return; }
    if (t[0] == 'f' && is_space(t[1])) {
        t += 2;
        t += strspn(t, " \t");
        if (k.face >= c.n_faces) { c.overflow = true; return; }
        if (k.mv != k.v || k.mvn != k.vn || k.mvt != k.vt) {
            Marker m; m.face = k.face; m.v = k.v; m.vn = k.vn; m.vt = k.vt;
            c.marks.push_back(m);
            k.mv = k.v; k.mvn = k.vn; k.mvt = k.vt;
        }
        o.face_first[c.base_face + k.face] = (uint32_t)(c.base_corner + k.corner);
        o.face_line[c.base_face + k.face] = k.line;
        while (!is_eol(t[0])) {
            // i, i/j, i//k, i/j/k
            const int vi = atoi(t);
            int vti = 0, vni = 0;
            t += strcspn(t, "/ \t\r");
            if (t[0] == '/') {
                t++;
                if (t[0] == '/') {
                    t++;
                    vni = atoi(t);
                    t += strcspn(t, "/ \t\r");
                } else {
                    vti = atoi(t);
                    t += strcspn(t, "/ \t\r");
                    if (t[0] == '/') {
                        t++;
                        vni = atoi(t);
                        t += strcspn(t, "/ \t\r");
                    }
                }
            }
            if (k.corner >= c.n_corners) { c.overflow = true; return; }
            o.corner[c.base_corner + k.corner] = vi;
            if (vti != 0) { if (c.vt.empty()) c.vt.assign(c.n_corners, 0); c.vt[k.corner] = vti; }
            if (vni != 0) { if (c.vn.empty()) c.vn.assign(c.n_corners, 0); c.vn[k.corner] = vni; }
            k.corner++;
            t += strspn(t, " \t\r");
        }
        k.face++;
        return;
    }
    Event ev;
    ev.face_before = k.face; ev.v_before = k.v; ev.vn_before = k.vn; ev.vt_before = k.vt; ev.text = 0;
    if (0 == strncmp(t, "usemtl", 6)) {
        t += 6;
        ev.kind = EV_USEMTL; ev.text = c.text.size();
        c.text.push_back(next_word(&t));
        c.events.push_back(ev);
        return;
    }
    if (0 == strncmp(t, "mtllib", 6) && is_space(t[6])) {
        ev.kind = EV_MTLLIB; ev.text = c.text.size();
        c.text.push_back(std::string(t + 7));
        c.events.push_back(ev);
        return;
    }
    if ((t[0] == 'g' || t[0] == 'o') && is_space(t[1])) {
        ev.kind = EV_GROUP;
        c.events.push_back(ev);
        return;
    }
    // everything else (vn/vt payloads, s, l, p, t ...) does not reach the wrapper's outputs
}

// Lines of the chunk: '\n' ends a line and is overwritten with NUL, trailing '\r's too (the chunk is ours to
// write; buf[c.end] is either the next chunk's first byte — never touched — or the spare NUL behind the file).
// A last line without '\n' counts if the chunk is not exhausted, like std::getline at EOF.
void tokenise_chunk(char* buf, Chunk& c, const Outputs& o)
{
    char* p = buf + c.begin;
    char* end = buf + c.end;
    ChunkCursor k;
    while (p < end) {
        char* nl = (char*)memchr(p, '\n', (size_t)(end - p));
        char* stop = nl ? nl : end;          // only the last chunk can lack the final '\n'; *end is then the spare NUL
        *stop = '\0';
        for (char* q = stop; q > p && (q[-1] == '\r' || q[-1] == '\n'); --q) q[-1] = '\0';
        k.line++;
        if (*p) tokenise_line(p, c, k, o);
        p = stop + 1;
    }
    if (k.v != c.n_v || k.face != c.n_faces || k.corner != c.n_corners) c.overflow = true;
}

size_t env_size(const char* name, size_t dflt)
{
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    long long v = atoll(e);
    return v > 0 ? (size_t)v : dflt;
}

template <typename F>
void run_parallel(size_t n_items, size_t threads, F&& fn)
{
    std::atomic<size_t> next(0);
    auto work = [&]() { for (;;) { const size_t k = next.fetch_add(1); if (k >= n_items) break; fn(k); } };
    const size_t n_thr = threads < n_items ? threads : n_items;
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_thr; t++) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
}

}  // namespace

TinyObjWrapper::TinyObjWrapper(const std::string& filename) { loadFile(filename); }

bool TinyObjWrapper::loadFile(const std::string& filename)
{
    _warn.clear(); _err.clear();
    _vertices.clear(); _materials.clear(); _materialIndices.clear(); _indexBuffer.clear();
    dataLoaded = false;
    const bool timing = getenv("ACGPT_OBJ_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };

    std::ifstream in(filename.c_str(), std::ios::binary);
    if (!in) {
        _err = "Cannot open file [" + filename + "]\n";
        std::cerr << "TinyObjReader: " << _err;
        return false;
    }
    std::string dir;
    size_t slash = filename.find_last_of("/\\");
    if (slash != std::string::npos) dir = filename.substr(0, slash);

    // whole file in memory, one spare byte for the final NUL
    std::vector<char> buf;
    {
        in.seekg(0, std::ios::end);
        std::streamoff len = in.tellg();
        in.seekg(0, std::ios::beg);
        if (len < 0) len = 0;
        buf.resize((size_t)len + 1);
        if (len > 0) in.read(buf.data(), len);
        buf.resize((size_t)in.gcount() + 1);
        buf.back() = '\0';
    }
    const size_t size = buf.size() - 1;
    const double t_read = since();

    // chunks at line ends.  ACGPT_OBJ_THREADS caps the threads (default: hardware threads, at most 16);
    // ACGPT_OBJ_CHUNK_BYTES is the smallest chunk worth a thread (default 1 MiB; tests shrink it).
    size_t threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 16) threads = 16;
    threads = env_size("ACGPT_OBJ_THREADS", threads);
    const size_t min_chunk = env_size("ACGPT_OBJ_CHUNK_BYTES", (size_t)1 << 20);
    size_t want = size / min_chunk;
    if (want < 1) want = 1;
    if (want > threads * 4) want = threads * 4;
    std::vector<size_t> cut;
    cut.push_back(0);
    for (size_t k = 1; k < want; k++) {
        size_t pos = size / want * k;
        if (pos <= cut.back()) continue;
        const char* nl = (const char*)memchr(buf.data() + pos, '\n', size - pos);
        if (!nl) break;
        pos = (size_t)(nl - buf.data()) + 1;
        if (pos > cut.back() && pos < size) cut.push_back(pos);
    }
    cut.push_back(size);
    std::vector<Chunk> chunks(cut.size() - 1);
    for (size_t k = 0; k < chunks.size(); k++) { chunks[k].begin = cut[k]; chunks[k].end = cut[k + 1]; }

    // ---- pass A: counts, then prefix sums ----------------------------------------------------------
    run_parallel(chunks.size(), threads, [&](size_t k) { count_chunk(buf.data(), chunks[k]); });
    size_t tot_v = 0, tot_vn = 0, tot_vt = 0, tot_line = 0, tot_face = 0, tot_corner = 0;
    for (Chunk& c : chunks) {
        c.base_v = tot_v; c.base_vn = tot_vn; c.base_vt = tot_vt; c.base_line = tot_line; c.base_face = tot_face; c.base_corner = tot_corner;
        tot_v += c.n_v; tot_vn += c.n_vn; tot_vt += c.n_vt; tot_line += c.n_lines; tot_face += c.n_faces; tot_corner += c.n_corners;
    }
    if (tot_corner >= 0xFFFFFFFFull || tot_v >= 0x7FFFFFFFull) {
        _err = "OBJ file too large for 32-bit indices\n";
        std::cerr << "TinyObjReader: " << _err;
        return false;
    }
    const double t_count = since();

    // ---- pass B: tokenise into place -----------------------------------------------------------------
    _vertices.resize(tot_v * 4);
    std::vector<int> corner(tot_corner);
    std::vector<uint32_t> face_first(tot_face + 1), face_line(tot_face);
    face_first[tot_face] = (uint32_t)tot_corner;
    Outputs outp;
    outp.vertices = _vertices.data(); outp.corner = corner.data(); outp.face_first = face_first.data(); outp.face_line = face_line.data();
    run_parallel(chunks.size(), threads, [&](size_t k) { tokenise_chunk(buf.data(), chunks[k], outp); });
    for (const Chunk& c : chunks)
        if (c.overflow) {
            _vertices.clear();
            _err = "internal error: OBJ statement counts changed between passes\n";
            std::cerr << "TinyObjReader: " << _err;
            return false;
        }
    const double t_tok = since();

    // ---- pass C: replay in file order ----------------------------------------------------------------
    std::vector<RawMaterial> raw;
    std::map<std::string, int> by_name;
    std::set<std::string> mtl_done;
    Mesh mesh;
    {
        long long tri_bound = (long long)tot_corner - 2 * (long long)tot_face;
        if (tri_bound < 0) tri_bound = 0;
        mesh.idx.reserve((size_t)tri_bound * 3);
        mesh.mat.reserve((size_t)tri_bound);
    }
    int material = -1;
    size_t pend_first = 0;          // faces [pend_first, done) wait for the next flush
    size_t done = 0;                // global ordinal of the next face to resolve
    bool ok = true;
    const float* V = _vertices.data();
    auto flush = [&](size_t n_v_now) {
        flush_faces(corner.data(), face_first.data(), pend_first, done, material, V, n_v_now, mesh, _warn);
        pend_first = done;
    };
    // resolve the faces of chunk c up to local ordinal `upto`
    auto resolve = [&](const Chunk& c, size_t& mark_i, size_t upto) -> bool {
        while (done < c.base_face + upto) {
            const size_t local = done - c.base_face;
            while (mark_i + 1 < c.marks.size() && c.marks[mark_i + 1].face <= local) mark_i++;
            const Marker& m = c.marks[mark_i];       // a face always has a marker at or before it
            const int n_v = (int)(c.base_v + m.v), n_vn = (int)(c.base_vn + m.vn), n_vt = (int)(c.base_vt + m.vt);
            for (uint32_t k = face_first[done]; k < face_first[done + 1]; k++) {
                int vi, dummy;
                if (!fix_index(corner[k], n_v, &vi)) return false;
                // vt before vn, and only non-zero values, as the line-by-line loader checks them
                const size_t lk = (size_t)k - c.base_corner;
                if (!c.vt.empty() && c.vt[lk] != 0 && !fix_index(c.vt[lk], n_vt, &dummy)) return false;
                if (!c.vn.empty() && c.vn[lk] != 0 && !fix_index(c.vn[lk], n_vn, &dummy)) return false;
                corner[k] = vi;
            }
            done++;
        }
        return true;
    };
    for (size_t ci = 0; ci < chunks.size() && ok; ci++) {
        const Chunk& c = chunks[ci];
        size_t mark_i = 0;
        for (size_t ei = 0; ei <= c.events.size() && ok; ei++) {
            const size_t upto = ei < c.events.size() ? c.events[ei].face_before : c.n_faces;
            if (!resolve(c, mark_i, upto)) {
                std::ostringstream ss;
                ss << "Failed to parse `f' line (e.g. a zero value for vertex index or invalid relative vertex index). Line " << (c.base_line + face_line[done]) << ").\n";
                _err += ss.str();
                ok = false;
                break;
            }
            if (ei == c.events.size()) break;
            const Event& ev = c.events[ei];
            const size_t n_v = c.base_v + ev.v_before;
            if (ev.kind == EV_USEMTL) {
                const std::string& name = c.text[ev.text];
                int id = -1;
                std::map<std::string, int>::const_iterator it = by_name.find(name);
                if (it != by_name.end()) id = it->second;
                else _warn += "material [ '" + name + "' ] not found in .mtl\n";
                if (id != material) {
                    flush(n_v);
                    material = id;
                }
            } else if (ev.kind == EV_MTLLIB) {
                // space separated list, backslash escapes a space; first file that opens wins
                std::vector<std::string> names;
                {
                    std::string cur; bool esc = false;
                    for (const char* p = c.text[ev.text].c_str(); *p; ++p) {
                        if (esc) { esc = false; }
                        else if (*p == '\\') { esc = true; continue; }
                        else if (*p == ' ') { if (!cur.empty()) names.push_back(cur); cur.clear(); continue; }
                        cur += *p;
                    }
                    names.push_back(cur);
                }
                bool found = false;
                for (const std::string& nm : names) {
                    if (mtl_done.count(nm)) { found = true; continue; }
                    std::string path = dir.empty() ? nm : (dir.back() == '/' ? dir + nm : dir + "/" + nm);
                    std::ifstream mf(path.c_str());
                    if (!mf) {
                        _warn += "Material file [ " + path + " ] not found in a path : " + dir + "\n";
                        continue;
                    }
                    load_mtl(mf, raw, by_name);
                    mtl_done.insert(nm);
                    found = true;
                    break;
                }
                if (!found) _warn += "Failed to load material file(s). Use default material.\n";
            } else {   // g / o: a new shape starts; order is file order
                flush(n_v);
            }
        }
    }
    if (ok) flush(tot_v);

    if (!ok) {
        _vertices.clear();
        if (!_err.empty()) std::cerr << "TinyObjReader: " << _err;
        if (!_warn.empty()) std::cout << "TinyObjReader: " << _warn;
        return false;
    }
    if (!_warn.empty()) std::cout << "TinyObjReader: " << _warn;

    for (const RawMaterial& m : raw) {
        Material mat;
        mat.diffuse = make_float3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
        mat.emission = make_float3(m.emission[0], m.emission[1], m.emission[2]);
        mat.roughness = m.roughness;
        mat.metallic = m.metallic;
        mat.ior = m.ior;
        if (m.name.find("Refractive") != std::string::npos) mat.bsdfType = BSDF_REFRACTION;
        else if (m.name.find("Metallic") != std::string::npos) mat.bsdfType = BSDF_METALLIC;
        else mat.bsdfType = BSDF_DIFFUSE;
        _materials.push_back(mat);
    }
    _materialIndices.swap(mesh.mat);
    _indexBuffer.swap(mesh.idx);
    dataLoaded = true;
    if (timing)
        std::cerr << "TinyObjWrapper: " << size << " bytes, " << chunks.size() << " chunks on " << (threads < chunks.size() ? threads : chunks.size())
                  << " threads: read " << t_read << " ms, count " << (t_count - t_read) << " ms, tokenise " << (t_tok - t_count)
                  << " ms, replay + triangulate " << (since() - t_tok) << " ms\n";
    return true;
}

std::vector<float> TinyObjWrapper::getVerticesFloat() const { return _vertices; }
std::vector<Material> TinyObjWrapper::getMaterials() const { return _materials; }
std::vector<uint32_t> TinyObjWrapper::getMaterialIndices() const { return _materialIndices; }
std::vector<uint32_t> TinyObjWrapper::getIndexBuffer() const { return _indexBuffer; }
size_t TinyObjWrapper::getNumMaterials() const { return _materials.size(); }

}  // namespace acgpt
