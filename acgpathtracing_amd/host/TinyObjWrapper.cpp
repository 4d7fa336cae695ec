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

struct Face { std::vector<int> v; };

// point-in-triangle by crossing number on the two projected axes
int pnpoly3(const float* vx, const float* vy, float tx, float ty)
{
    int c = 0;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
    return c;
}

struct Mesh { std::vector<uint32_t> idx; std::vector<uint32_t> mat; };

void emit_tri(Mesh& m, int a, int b, int c, int mat)
{
    m.idx.push_back((uint32_t)a); m.idx.push_back((uint32_t)b); m.idx.push_back((uint32_t)c);
    m.mat.push_back((uint32_t)mat);
}

// Triangulate the pending faces against the vertex array as it stands now (tinyobj
// triangulates when a group is flushed, not when the `f` line is read).
void flush_faces(std::vector<Face>& faces, int mat, const std::vector<float>& v, Mesh& out, std::string& warn)
{
    for (const Face& f : faces) {
        size_t n = f.v.size();
        if (n < 3) { warn += "Degenerated face found\n."; continue; }
        if (n == 3) { emit_tri(out, f.v[0], f.v[1], f.v[2], mat); continue; }
        if (n == 4) {
            size_t a = (size_t)f.v[0], b = (size_t)f.v[1], c = (size_t)f.v[2], d = (size_t)f.v[3];
            if (3 * a + 2 >= v.size() || 3 * b + 2 >= v.size() || 3 * c + 2 >= v.size() || 3 * d + 2 >= v.size()) {
                warn += "Face with invalid vertex index found.\n";
                continue;
            }
            // split along the shorter diagonal: |v2-v0|^2 < |v3-v1|^2 -> (0,1,2)(0,2,3), else (0,1,3)(1,2,3)
            float acx = v[3 * c] - v[3 * a], acy = v[3 * c + 1] - v[3 * a + 1], acz = v[3 * c + 2] - v[3 * a + 2];
            float bdx = v[3 * d] - v[3 * b], bdy = v[3 * d + 1] - v[3 * b + 1], bdz = v[3 * d + 2] - v[3 * b + 2];
            float d02 = acx * acx + acy * acy + acz * acz;
            float d13 = bdx * bdx + bdy * bdy + bdz * bdz;
            if (d02 < d13) { emit_tri(out, f.v[0], f.v[1], f.v[2], mat); emit_tri(out, f.v[0], f.v[2], f.v[3], mat); }
            else           { emit_tri(out, f.v[0], f.v[1], f.v[3], mat); emit_tri(out, f.v[1], f.v[2], f.v[3], mat); }
            continue;
        }
        // n > 4: ear clipping in the plane of the first non-degenerate corner
        size_t ax0 = 1, ax1 = 2;
        for (size_t k = 0; k < n; k++) {
            size_t i0 = (size_t)f.v[k % n], i1 = (size_t)f.v[(k + 1) % n], i2 = (size_t)f.v[(k + 2) % n];
            if (3 * i0 + 2 >= v.size() || 3 * i1 + 2 >= v.size() || 3 * i2 + 2 >= v.size()) continue;
            float e0x = v[3 * i1] - v[3 * i0], e0y = v[3 * i1 + 1] - v[3 * i0 + 1], e0z = v[3 * i1 + 2] - v[3 * i0 + 2];
            float e1x = v[3 * i2] - v[3 * i1], e1y = v[3 * i2 + 1] - v[3 * i1 + 1], e1z = v[3 * i2 + 2] - v[3 * i1 + 2];
            float cx = std::fabs(e0y * e1z - e0z * e1y);
            float cy = std::fabs(e0z * e1x - e0x * e1z);
            float cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = std::numeric_limits<float>::epsilon();
            if (cx > eps || cy > eps || cz > eps) {
                if (!(cx > cy && cx > cz)) { ax0 = 0; if (cz > cx && cz > cy) ax1 = 1; }
                break;
            }
        }
        std::vector<int> rem = f.v;
        size_t guess = 0;
        size_t budget = f.v.size();
        size_t prev_n = rem.size();
        while (rem.size() > 3 && budget > 0) {
            size_t m = rem.size();
            if (guess >= m) guess -= m;
            if (prev_n != m) { prev_n = m; budget = m; } else { budget--; }
            int ind[3]; float vx[3], vy[3];
            for (size_t k = 0; k < 3; k++) {
                ind[k] = rem[(guess + k) % m];
                size_t vi = (size_t)ind[k];
                if (vi * 3 + ax0 >= v.size() || vi * 3 + ax1 >= v.size()) { vx[k] = 0.0f; vy[k] = 0.0f; }
                else { vx[k] = v[vi * 3 + ax0]; vy[k] = v[vi * 3 + ax1]; }
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
                if (ovi * 3 + ax0 >= v.size() || ovi * 3 + ax1 >= v.size()) continue;
                if (pnpoly3(vx, vy, v[ovi * 3 + ax0], v[ovi * 3 + ax1])) { overlap = true; break; }
            }
            if (overlap) { guess += 1; continue; }
            emit_tri(out, ind[0], ind[1], ind[2], mat);             // an ear: cut its middle vertex
            size_t gone = (guess + 1) % m;
            rem.erase(rem.begin() + (long)gone);
        }
        if (rem.size() == 3) emit_tri(out, rem[0], rem[1], rem[2], mat);
    }
    faces.clear();
}

}  // namespace

TinyObjWrapper::TinyObjWrapper(const std::string& filename) { loadFile(filename); }

bool TinyObjWrapper::loadFile(const std::string& filename)
{
    _warn.clear(); _err.clear();
    _vertices.clear(); _materials.clear(); _materialIndices.clear(); _indexBuffer.clear();
    dataLoaded = false;

    std::ifstream in(filename.c_str());
    if (!in) {
        _err = "Cannot open file [" + filename + "]\n";
        std::cerr << "TinyObjReader: " << _err;
        return false;
    }
    std::string dir;
    size_t slash = filename.find_last_of("/\\");
    if (slash != std::string::npos) dir = filename.substr(0, slash);

    std::vector<float> v;
    std::vector<RawMaterial> raw;
    std::map<std::string, int> by_name;
    std::set<std::string> mtl_done;
    std::vector<Face> pending;
    Mesh mesh;
    int material = -1;
    int n_vn = 0, n_vt = 0;
    size_t line_no = 0;
    bool ok = true;
    std::string line;
    while (get_line(in, line)) {
        line_no++;
        if (line.empty()) continue;
        const char* t = line.c_str();
        t += strspn(t, " \t");
        if (t[0] == '\0' || t[0] == '#') continue;

        if (t[0] == 'v' && is_space(t[1])) {
            t += 2;
            float x = next_real(&t), y = next_real(&t), z = next_real(&t);
            v.push_back(x); v.push_back(y); v.push_back(z);
            continue;
        }
        if (t[0] == 'v' && t[1] == 'n' && is_space(t[2])) { n_vn++; continue; }
        if (t[0] == 'v' && t[1] == 't' && is_space(t[2])) { n_vt++; continue; }
        if (t[0] == 'f' && is_space(t[1])) {
            t += 2;
            t += strspn(t, " \t");
            Face f;
            while (!is_eol(t[0])) {
                // i, i/j, i//k, i/j/k : only i matters here, j and k are validated like tinyobj does
                int vi;
                if (!fix_index(atoi(t), (int)(v.size() / 3), &vi)) { ok = false; break; }
                t += strcspn(t, "/ \t\r");
                if (t[0] == '/') {
                    t++;
                    int dummy;
                    if (t[0] == '/') {
                        t++;
                        if (atoi(t) != 0 && !fix_index(atoi(t), n_vn, &dummy)) { ok = false; break; }
                        t += strcspn(t, "/ \t\r");
                    } else {
                        if (atoi(t) != 0 && !fix_index(atoi(t), n_vt, &dummy)) { ok = false; break; }
                        t += strcspn(t, "/ \t\r");
                        if (t[0] == '/') {
                            t++;
                            if (atoi(t) != 0 && !fix_index(atoi(t), n_vn, &dummy)) { ok = false; break; }
                            t += strcspn(t, "/ \t\r");
                        }
                    }
                }
                f.v.push_back(vi);
                t += strspn(t, " \t\r");
            }
            if (!ok) {
                std::ostringstream ss;
                ss << "Failed to parse `f' line (e.g. a zero value for vertex index or invalid relative vertex index). Line " << line_no << ").\n";
                _err += ss.str();
                break;
            }
            pending.push_back(f);
            continue;
        }
        if (0 == strncmp(t, "usemtl", 6)) {
            t += 6;
            std::string name = next_word(&t);
            int id = -1;
            std::map<std::string, int>::const_iterator it = by_name.find(name);
            if (it != by_name.end()) id = it->second;
            else _warn += "material [ '" + name + "' ] not found in .mtl\n";
            if (id != material) {
                flush_faces(pending, material, v, mesh, _warn);
                material = id;
            }
            continue;
        }
        if (0 == strncmp(t, "mtllib", 6) && is_space(t[6])) {
            t += 7;
            // space separated list, backslash escapes a space; first file that opens wins
            std::vector<std::string> names;
            {
                std::string cur; bool esc = false;
                for (const char* p = t; *p; ++p) {
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
            continue;
        }
        if ((t[0] == 'g' || t[0] == 'o') && is_space(t[1])) {
            flush_faces(pending, material, v, mesh, _warn);   // a new shape starts; order is file order
            continue;
        }
        // everything else (vn/vt payloads, s, l, p, t ...) does not reach the wrapper's outputs
    }
    if (ok) flush_faces(pending, material, v, mesh, _warn);

    if (!ok) {
        if (!_err.empty()) std::cerr << "TinyObjReader: " << _err;
        if (!_warn.empty()) std::cout << "TinyObjReader: " << _warn;
        return false;
    }
    if (!_warn.empty()) std::cout << "TinyObjReader: " << _warn;

    _vertices.reserve(v.size() / 3 * 4);
    for (size_t i = 0; i + 2 < v.size(); i += 3) {
        _vertices.push_back(v[i]); _vertices.push_back(v[i + 1]); _vertices.push_back(v[i + 2]); _vertices.push_back(1.0f);
    }
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
    return true;
}

std::vector<float> TinyObjWrapper::getVerticesFloat() const { return _vertices; }
std::vector<Material> TinyObjWrapper::getMaterials() const { return _materials; }
std::vector<uint32_t> TinyObjWrapper::getMaterialIndices() const { return _materialIndices; }
std::vector<uint32_t> TinyObjWrapper::getIndexBuffer() const { return _indexBuffer; }
size_t TinyObjWrapper::getNumMaterials() const { return _materials.size(); }

}  // namespace acgpt
