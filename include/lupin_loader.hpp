// lupin_loader.hpp -- C++ counterpart of the reference's `lupin_loader` crate for the formats its test scenes use
// ("next" row 8f-2): Yocto/GL 2.4 JSON scenes, binary little-endian PLY meshes, PNG and Radiance HDR textures.
//
//   lpl::load_scene_cpu_yoctogl_v24 / load_scene_yoctogl_v24   lupin_loader/src/loader.rs:331-768 (+ materials :770-911)
//   lpl::load_mesh_ply                                         loader.rs:1274-1566
//   lpl::load_texture                                          loader.rs:214-293 (HDR -> Rgba16Float, others -> Rgba8Unorm)
//
// It fills lp::SceneCPU / lp::TextureCPU / lp::EnvMapInfo of include/lupin.hpp and hands them to
// lp::build_accel_structures_and_upload, exactly as the Python host (lupinpathtracer_amd/loader.py) does; the two are
// tested to produce byte-identical scene arrays.  Header-only; link with -llupin_hip -lz (PNG inflate).
// Where the reference returns Err / panics these functions throw lpl::LoadError.
#pragma once

#include <zlib.h>

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <utility>

#include "lupin.hpp"

namespace lpl {

struct LoadError : std::runtime_error { using std::runtime_error::runtime_error; };

inline std::vector<uint8_t> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw LoadError("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// ------------------------------------------------------------------------------------------------
// JSON (just what a scene file needs; object members keep FILE ORDER, which the reference's parser depends on:
// "color" resets opacity, environments inherit fields from the previous entry -- loader.rs:444-445,791-792)
// ------------------------------------------------------------------------------------------------
struct Json
{
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json *find(const std::string &key) const
    {
        for (const auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

class JsonParser
{
  public:
    explicit JsonParser(const std::string &text) : s(text) {}
    Json parse()
    {
        Json v = value();
        ws();
        if (p != s.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s;
    size_t p = 0;
    [[noreturn]] void fail(const char *what) const { throw LoadError(std::string("JSON: ") + what + " at byte " + std::to_string(p)); }
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++; }
    bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { p++; return true; } return false; }
    Json value()
    {
        ws();
        if (p >= s.size()) fail("unexpected end");
        Json v;
        const char c = s[p];
        if (c == '{')
        {
            p++; v.kind = Json::Object;
            if (eat('}')) return v;
            do
            {
                ws();
                if (p >= s.size() || s[p] != '"') fail("expected a member name");
                std::string key = string();
                if (!eat(':')) fail("expected ':'");
                v.obj.emplace_back(std::move(key), value());
            } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        }
        else if (c == '[')
        {
            p++; v.kind = Json::Array;
            if (eat(']')) return v;
            do v.arr.push_back(value()); while (eat(','));
            if (!eat(']')) fail("expected ']'");
        }
        else if (c == '"') { v.kind = Json::String; v.str = string(); }
        else if (s.compare(p, 4, "true") == 0) { v.kind = Json::Bool; v.b = true; p += 4; }
        else if (s.compare(p, 5, "false") == 0) { v.kind = Json::Bool; v.b = false; p += 5; }
        else if (s.compare(p, 4, "null") == 0) { p += 4; }
        else
        {
            const char *begin = s.c_str() + p;
            char *end = nullptr;
            v.num = std::strtod(begin, &end);
            if (end == begin) fail("unexpected character");
            v.kind = Json::Number;
            p += (size_t)(end - begin);
        }
        return v;
    }
    std::string string()
    {
        std::string out;
        p++;   // opening quote
        while (p < s.size() && s[p] != '"')
        {
            char c = s[p++];
            if (c == '\\' && p < s.size())
            {
                const char e = s[p++];
                switch (e)
                {
                case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break;
                case 'b': c = '\b'; break; case 'f': c = '\f'; break;
                case 'u':
                {
                    if (p + 4 > s.size()) fail("bad \\u escape");
                    const unsigned cp = (unsigned)std::strtoul(s.substr(p, 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) c = (char)cp;
                    else { out += (char)(0xC0 | (cp >> 6)); c = (char)(0x80 | (cp & 0x3F)); }   // scene files are ASCII; 2-byte UTF-8 suffices
                    break;
                }
                default: c = e; break;   // \" \\ \/
                }
            }
            out += c;
        }
        if (p >= s.size()) fail("unterminated string");
        p++;
        return out;
    }
};

// ------------------------------------------------------------------------------------------------
// Images
// ------------------------------------------------------------------------------------------------

// IEEE f32 -> f16, round to nearest even (half::f16::from_f32, loader.rs:244-247)
inline uint16_t float_to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u | ((x >> 13) & 0x3FFu) : 0u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                 // rounds to infinity
    if (x < 0x33000001u) return (uint16_t)sign;                              // rounds to zero
    int exp = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t base;
    if (exp < -14) { shift = 13 + (-14 - exp); base = 0; }                   // subnormal half
    else { shift = 13; base = (uint32_t)(exp + 15) << 10; man &= 0x7FFFFFu; }
    uint32_t h = base + (man >> shift);
    const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

// Radiance RGBE -> w * h * 3 floats.  Decoding rule of the `image` crate: mantissa * 2^(e - 136), e == 0 -> 0.
inline std::vector<float> read_hdr(const std::string &path, uint32_t &w, uint32_t &h)
{
    const std::vector<uint8_t> d = read_file(path);
    if (d.size() < 2 || d[0] != '#' || d[1] != '?') throw LoadError(path + ": not a Radiance file");
    size_t pos = 0;
    auto line = [&]() {
        size_t end = pos;
        while (end < d.size() && d[end] != '\n') end++;
        if (end >= d.size()) throw LoadError(path + ": truncated header");
        std::string l((const char *)&d[pos], end - pos);
        pos = end + 1;
        return l;
    };
    for (;;)
    {
        std::string l = line();
        if (l.find_first_not_of(" \t\r") == std::string::npos) break;
    }
    std::istringstream res(line());
    std::string a, c;
    long hh = 0, ww = 0;
    res >> a >> hh >> c >> ww;
    if (a != "-Y" || c != "+X" || hh <= 0 || ww <= 0) throw LoadError(path + ": unsupported orientation");
    w = (uint32_t)ww; h = (uint32_t)hh;
    std::vector<uint8_t> rgbe((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++)
    {
        uint8_t *row = &rgbe[(size_t)y * w * 4];
        if (pos + 4 > d.size()) throw LoadError(path + ": truncated");
        if (w < 8 || w > 0x7FFF || !(d[pos] == 2 && d[pos + 1] == 2 && (d[pos + 2] & 0x80) == 0))
        {
            if (pos + (size_t)4 * w > d.size()) throw LoadError(path + ": truncated");
            std::memcpy(row, &d[pos], (size_t)4 * w);
            pos += (size_t)4 * w;
            continue;
        }
        if ((((uint32_t)d[pos + 2] << 8) | d[pos + 3]) != w) throw LoadError(path + ": bad scanline width");
        pos += 4;
        for (int ch = 0; ch < 4; ch++)
            for (uint32_t x = 0; x < w;)
            {
                if (pos >= d.size()) throw LoadError(path + ": truncated");
                uint32_t n = d[pos++];
                if (n > 128)
                {
                    n -= 128;
                    if (x + n > w || pos >= d.size()) throw LoadError(path + ": bad run");
                    for (uint32_t k = 0; k < n; k++) row[(size_t)(x + k) * 4 + ch] = d[pos];
                    pos++;
                }
                else
                {
                    if (x + n > w || pos + n > d.size()) throw LoadError(path + ": bad run");
                    for (uint32_t k = 0; k < n; k++) row[(size_t)(x + k) * 4 + ch] = d[pos + k];
                    pos += n;
                }
                x += n;
            }
    }
    std::vector<float> out((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++)
    {
        const int e = rgbe[i * 4 + 3];
        const float scale = e == 0 ? 0.0f : std::ldexp(1.0f, e - 136);
        for (int k = 0; k < 3; k++) out[i * 3 + k] = (float)rgbe[i * 4 + k] * scale;
    }
    return out;
}

// PNG -> RGBA8 (8-bit greyscale / RGB / palette / grey+alpha / RGBA, non-interlaced; what `image::open(..).to_rgba8()`
// yields for such files)
inline std::vector<uint8_t> read_png_rgba8(const std::string &path, uint32_t &w, uint32_t &h)
{
    const std::vector<uint8_t> d = read_file(path);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) throw LoadError(path + ": not a PNG file");
    auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    for (size_t p = 8; p + 12 <= d.size();)
    {
        const uint32_t len = be32(p);
        const std::string type((const char *)&d[p + 4], 4);
        if (p + 12 + (size_t)len > d.size()) throw LoadError(path + ": truncated chunk");
        const uint8_t *body = &d[p + 8];
        if (type == "IHDR") { w = be32(p + 8); h = be32(p + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "tRNS") trns.assign(body, body + len);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        p += 12 + (size_t)len;
    }
    if (depth != 8 || interlace != 0) throw LoadError(path + ": only 8-bit non-interlaced PNGs are supported");
    const uint32_t ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || !w || !h) throw LoadError(path + ": unsupported PNG colour type");
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) throw LoadError(path + ": inflate failed");
    std::vector<uint8_t> px(stride * h);
    for (uint32_t y = 0; y < h; y++)
    {
        const uint8_t filter = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &px[stride * y];
        const uint8_t *up = y ? &px[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++)
        {
            const int a = i >= ch ? dst[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= ch) ? up[i - ch] : 0;
            int pred = 0;
            switch (filter)
            {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: throw LoadError(path + ": bad PNG filter");
            }
            dst[i] = (uint8_t)(src[i] + pred);
        }
    }
    std::vector<uint8_t> out((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++)
    {
        uint8_t r, g, b, a = 255;
        const uint8_t *s = &px[i * ch];
        if (ctype == 0) { r = g = b = s[0]; }
        else if (ctype == 2) { r = s[0]; g = s[1]; b = s[2]; }
        else if (ctype == 3)
        {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) throw LoadError(path + ": palette index out of range");
            r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2];
            if (s[0] < trns.size()) a = trns[s[0]];
        }
        else if (ctype == 4) { r = g = b = s[0]; a = s[1]; }
        else { r = s[0]; g = s[1]; b = s[2]; a = s[3]; }
        out[i * 4] = r; out[i * 4 + 1] = g; out[i * 4 + 2] = b; out[i * 4 + 3] = a;
    }
    return out;
}

// load_texture_with_usage (loader.rs:214-286): `texture` is what goes to the device, `f32` (RGBA) feeds the env alias table
struct LoadedTexture { lp::TextureCPU texture; std::vector<lp::Vec4> f32; };
inline LoadedTexture load_texture(const std::string &path)
{
    LoadedTexture out;
    std::string ext = path.substr(path.find_last_of('.') == std::string::npos ? path.size() : path.find_last_of('.') + 1);
    for (char &c : ext) c = (char)std::tolower((unsigned char)c);
    uint32_t w = 0, h = 0;
    if (ext == "hdr")
    {
        const std::vector<float> rgb = read_hdr(path, w, h);
        out.texture.format = LUPIN_TEX_RGBA16_FLOAT;
        out.texture.pixels.resize((size_t)w * h * 8);
        out.f32.resize((size_t)w * h);
        uint16_t *hp = reinterpret_cast<uint16_t *>(out.texture.pixels.data());
        for (size_t i = 0; i < (size_t)w * h; i++)
        {
            out.f32[i] = lp::Vec4{rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2], 1.0f};
            hp[i * 4] = float_to_half(rgb[i * 3]); hp[i * 4 + 1] = float_to_half(rgb[i * 3 + 1]); hp[i * 4 + 2] = float_to_half(rgb[i * 3 + 2]);
            hp[i * 4 + 3] = 0x3C00;
        }
    }
    else if (ext == "png")
    {
        out.texture.format = LUPIN_TEX_RGBA8_UNORM;
        out.texture.pixels = read_png_rgba8(path, w, h);
        out.f32.resize((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; i++)
        {
            const uint8_t *p = &out.texture.pixels[i * 4];
            out.f32[i] = lp::Vec4{(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
        }
    }
    else throw LoadError(path + ": unsupported texture format (png and hdr are)");
    out.texture.width = w; out.texture.height = h;
    return out;
}

// ------------------------------------------------------------------------------------------------
// PLY (load_mesh_ply, loader.rs:1274-1566): binary little-endian, float vertex properties, uchar-count face lists,
// polygons fan-triangulated (:1544-1562), V flipped (:1431-1435).  Returns the new mesh index.
// ------------------------------------------------------------------------------------------------
inline uint32_t load_mesh_ply(const std::string &path, lp::SceneCPU &scene)
{
    const std::vector<uint8_t> d = read_file(path);
    const std::string text((const char *)d.data(), std::min<size_t>(d.size(), 1 << 16));
    const size_t eh = text.find("end_header");
    if (eh == std::string::npos || text.compare(0, 3, "ply") != 0) throw LoadError(path + ": invalid PLY");
    const size_t header_end = text.find('\n', eh) + 1;
    std::istringstream hdr(text.substr(0, header_end));
    std::string line, section;
    std::map<std::string, size_t> offsets;
    size_t offset = 0, num_verts = 0, num_faces = 0;
    std::getline(hdr, line);
    while (std::getline(hdr, line))
    {
        std::istringstream ls(line);
        std::vector<std::string> tok;
        for (std::string t; ls >> t;) tok.push_back(t);
        if (tok.empty() || tok[0] == "comment") continue;
        if (tok[0] == "format")
        {
            if (tok.size() < 3 || tok[1] != "binary_little_endian" || tok[2] != "1.0") throw LoadError(path + ": only binary_little_endian 1.0 is supported");
        }
        else if (tok[0] == "element" && tok.size() >= 3)
        {
            section = tok[1];
            if (section == "vertex") num_verts = (size_t)std::stoull(tok[2]);
            else if (section == "face") num_faces = (size_t)std::stoull(tok[2]);
        }
        else if (tok[0] == "property" && tok.size() >= 3)
        {
            if (section == "vertex")
            {
                std::string name = tok[2] == "s" ? "u" : tok[2] == "t" ? "v" : tok[2];
                offsets[name] = offset;
                offset += tok[1] == "float" ? 4 : 0;   // only `float` properties occupy space in the reference's reader (:1337-1342)
            }
            else if (section == "face")
            {
                if (tok.size() < 4 || tok[1] != "list" || tok[2] != "uchar" || (tok[3] != "uint" && tok[3] != "int")) throw LoadError(path + ": unsupported face property");
            }
        }
    }
    const size_t stride = offset;
    if (!offsets.count("x") || !offsets.count("y") || !offsets.count("z")) throw LoadError(path + ": missing positions");
    if (header_end + num_verts * stride > d.size()) throw LoadError(path + ": truncated vertex data");
    auto f32_at = [&](size_t v, size_t off) { float f; std::memcpy(&f, &d[header_end + v * stride + off], 4); return f; };
    auto has = [&](const char *k) { return offsets.count(k) != 0; };
    auto col = [&](size_t v, const char *k) { return has(k) ? f32_at(v, offsets[k]) : 0.0f; };

    lp::MeshInfo info = lp::default_mesh_info();
    std::vector<lp::Vec4> pos(num_verts);
    for (size_t v = 0; v < num_verts; v++) pos[v] = lp::Vec4{col(v, "x"), col(v, "y"), col(v, "z"), 0.0f};
    if (has("nx") || has("ny") || has("nz"))
    {
        std::vector<lp::Vec4> n(num_verts);
        for (size_t v = 0; v < num_verts; v++) n[v] = lp::Vec4{col(v, "nx"), col(v, "ny"), col(v, "nz"), 0.0f};
        scene.verts_normal_array.push_back(std::move(n));
        info.normals_buf_idx = (uint32_t)scene.verts_normal_array.size() - 1;
    }
    if (has("u") || has("v"))
    {
        std::vector<float> uv(num_verts * 2);
        for (size_t v = 0; v < num_verts; v++) { uv[2 * v] = col(v, "u"); uv[2 * v + 1] = 1.0f - col(v, "v"); }
        scene.verts_texcoord_array.push_back(std::move(uv));
        info.texcoords_buf_idx = (uint32_t)scene.verts_texcoord_array.size() - 1;
    }
    if (has("red") || has("green") || has("blue") || has("alpha"))
    {
        std::vector<lp::Vec4> c(num_verts);
        for (size_t v = 0; v < num_verts; v++) c[v] = lp::Vec4{col(v, "red"), col(v, "green"), col(v, "blue"), col(v, "alpha")};
        scene.verts_color_array.push_back(std::move(c));
        info.colors_buf_idx = (uint32_t)scene.verts_color_array.size() - 1;
    }
    std::vector<uint32_t> indices;
    indices.reserve(num_faces * 3);
    size_t p = header_end + num_verts * stride;
    for (size_t f = 0; f < num_faces; f++)
    {
        if (p >= d.size()) throw LoadError(path + ": truncated face data");
        const uint32_t n = d[p++];
        if (p + (size_t)4 * n > d.size()) throw LoadError(path + ": truncated face data");
        auto id = [&](uint32_t k) { uint32_t v; std::memcpy(&v, &d[p + (size_t)4 * k], 4); return v; };
        for (uint32_t j = 1; j + 1 < n; j++) { indices.push_back(id(0)); indices.push_back(id(j)); indices.push_back(id(j + 1)); }
        p += (size_t)4 * n;
    }
    for (uint32_t i : indices) if (i >= num_verts) throw LoadError(path + ": vertex index out of range");
    scene.mesh_infos.push_back(info);
    scene.verts_pos_array.push_back(std::move(pos));
    scene.indices_array.push_back(std::move(indices));
    return (uint32_t)scene.mesh_infos.size() - 1;
}

// ------------------------------------------------------------------------------------------------
// Yocto/GL 2.4 JSON (load_scene_yoctogl_v24, loader.rs:331-768; parse_material_yocto_v24, :770-911)
// ------------------------------------------------------------------------------------------------
namespace detail {

inline lp::Mat3x4 parse_mat3x4(const Json &v)   // parse_mat3x4f (loader.rs:1074-1097): 12 numbers, column by column
{
    if (v.kind != Json::Array || v.arr.size() != 12) throw LoadError("frame must have 12 numbers");
    lp::Mat3x4 m{};
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) m.m[c][r] = (float)v.arr[(size_t)c * 3 + r].num;
    return m;
}

inline lp::Mat3x4 mul3x4(const lp::Mat3x4 &a, const lp::Mat3x4 &b)   // Mat3x4 * Mat3x4 (base.rs:738-757), f32, k = 0..3 in order
{
    float a4[4][4] = {}, b4[4][4] = {};
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) { a4[c][r] = a.m[c][r]; b4[c][r] = b.m[c][r]; }
    a4[3][3] = b4[3][3] = 1.0f;
    lp::Mat3x4 res{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
        {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc = acc + a4[k][i] * b4[j][k];
            res.m[j][i] = acc;
        }
    return res;
}

inline lp::Instance instance_from_transform(const lp::Mat3x4 &local_to_world, uint32_t mesh_idx, uint32_t mat_idx)   // loader.rs:653-654
{
    lp::Mat3x4 inv{};
    lupin_mat3x4_inverse(&local_to_world, &inv);
    lp::Instance in = lp::default_instance();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) in.transpose_inverse_transform.m[r][c] = inv.m[c][r];   // Mat3x4::transpose
    in.mesh_idx = mesh_idx; in.mat_idx = mat_idx;
    return in;
}

inline float num(const Json &v) { if (v.kind != Json::Number) throw LoadError("expected a number"); return (float)v.num; }
inline void vec3(const Json &v, float *out) { if (v.kind != Json::Array || v.arr.size() < 3) throw LoadError("expected 3 numbers"); for (int k = 0; k < 3; k++) out[k] = num(v.arr[k]); }
inline uint32_t idx(const Json &v) { if (v.kind != Json::Number) throw LoadError("expected an index"); return (uint32_t)(int64_t)v.num; }

inline lp::Material parse_material(const Json &d)
{
    static const std::map<std::string, uint32_t> types = {{"matte", 0}, {"glossy", 1}, {"reflective", 2}, {"transparent", 3}, {"refractive", 4},
                                                           {"subsurface", 5}, {"volume", 6}, {"gltfpbr", 7}};
    lp::Material m = lp::default_material();
    for (const auto &kv : d.obj)   // file order matters: "color" resets opacity to 1 (loader.rs:791-792)
    {
        const std::string &key = kv.first;
        const Json &val = kv.second;
        if (key == "color") { vec3(val, m.color); m.color[3] = 1.0f; }
        else if (key == "emission") vec3(val, m.emission);
        else if (key == "scattering") vec3(val, m.scattering);
        else if (key == "roughness") m.roughness = num(val);
        else if (key == "metallic") m.metallic = num(val);
        else if (key == "ior") m.ior = num(val);
        else if (key == "scanisotropy") m.sc_anisotropy = num(val);
        else if (key == "trdepth") m.tr_depth = num(val);
        else if (key == "opacity") m.color[3] = num(val);
        else if (key == "type") { auto it = types.find(val.str); if (it != types.end()) m.mat_type = it->second; }
        else if (key == "color_tex") m.color_tex_idx = idx(val);
        else if (key == "emission_tex") m.emission_tex_idx = idx(val);
        else if (key == "roughness_tex") m.roughness_tex_idx = idx(val);
        else if (key == "scattering_tex") m.scattering_tex_idx = idx(val);
        else if (key == "normal_tex") m.normal_tex_idx = idx(val);
    }
    return m;
}

inline std::string find_asset(const std::string &rel, const std::vector<std::string> &dirs)
{
    for (const std::string &dir : dirs)
    {
        const std::string p = dir + "/" + rel;
        if (std::ifstream(p).good()) return p;
    }
    throw LoadError("asset " + rel + " not found");
}

}  // namespace detail

struct LoadedSceneCPU
{
    lp::SceneCPU scene;
    std::vector<lp::TextureCPU> textures;
    std::vector<lp::EnvMapInfo> envs_info;
    std::vector<SceneCamera> cameras;
};

// Parse a Yocto/GL 2.4 scene without touching a device.  Assets are looked up next to the JSON, then in `asset_dirs`.
inline LoadedSceneCPU load_scene_cpu_yoctogl_v24(const std::string &path, const std::vector<std::string> &asset_dirs = {})
{
    using namespace detail;
    const std::vector<uint8_t> bytes = read_file(path);
    const std::string text(bytes.begin(), bytes.end());
    const Json doc = JsonParser(text).parse();
    if (doc.kind != Json::Object) throw LoadError(path + ": the top level must be an object");
    std::vector<std::string> dirs;
    const size_t slash = path.find_last_of('/');
    dirs.push_back(slash == std::string::npos ? std::string(".") : path.substr(0, slash));
    dirs.insert(dirs.end(), asset_dirs.begin(), asset_dirs.end());

    lp::Mat3x4 conversion = lp::mat3x4_identity();
    conversion.m[2][2] = -1.0f;   // Z flip into Lupin's left-handed frame (loader.rs:345-349)

    LoadedSceneCPU out;
    lp::SceneCPU &scene = out.scene;
    std::vector<std::string> tex_paths;
    uint32_t tex_referenced = 0;
    auto note_tex = [&](uint32_t i) { if (i != lp::SENTINEL_IDX) tex_referenced = std::max(tex_referenced, i + 1); };

    for (const auto &section : doc.obj)
    {
        const Json &items = section.second;
        if (section.first == "cameras")
        {
            for (const Json &c : items.arr)
            {
                SceneCamera cam;
                for (const auto &kv : c.obj)
                {
                    if (kv.first == "aspect") cam.params.aspect = num(kv.second);
                    else if (kv.first == "focus") cam.params.focus = num(kv.second);
                    else if (kv.first == "aperture") cam.params.aperture = num(kv.second);
                    else if (kv.first == "lens") cam.params.lens = num(kv.second);
                    else if (kv.first == "film") cam.params.film = num(kv.second);
                    else if (kv.first == "orthographic") cam.params.is_orthographic = kv.second.kind == Json::Bool ? kv.second.b : kv.second.num != 0.0;
                    else if (kv.first == "frame") cam.transform = mul3x4(mul3x4(conversion, parse_mat3x4(kv.second)), conversion);
                }
                out.cameras.push_back(cam);
            }
        }
        else if (section.first == "environments")
        {
            lp::Environment env{};
            env.emission_tex_idx = lp::SENTINEL_IDX;
            for (int k = 0; k < 4; k++) env.transform.m[k][k] = 1.0f;
            env.transform.m[2][2] = -1.0f;   // conversion_mat4 * IDENTITY
            for (const Json &e : items.arr)   // `env` is declared outside the loop in the reference: fields persist (loader.rs:444-445)
            {
                for (const auto &kv : e.obj)
                {
                    if (kv.first == "emission") vec3(kv.second, env.emission);
                    else if (kv.first == "emission_tex") { env.emission_tex_idx = idx(kv.second); note_tex(env.emission_tex_idx); }
                    else if (kv.first == "frame")
                    {
                        const lp::Mat3x4 fm = parse_mat3x4(kv.second);
                        for (int c = 0; c < 4; c++) { for (int r = 0; r < 3; r++) env.transform.m[c][r] = fm.m[c][r]; env.transform.m[c][3] = c == 3 ? 1.0f : 0.0f; }
                        for (int c = 0; c < 4; c++) env.transform.m[c][2] *= -1.0f;   // conversion_mat4 * m negates z of every column
                    }
                }
                scene.environments.push_back(env);
            }
        }
        else if (section.first == "textures")
        {
            for (const Json &t : items.arr) { const Json *uri = t.find("uri"); tex_paths.push_back(uri ? uri->str : std::string()); }
        }
        else if (section.first == "materials")
        {
            for (const Json &m : items.arr)
            {
                scene.materials.push_back(parse_material(m));
                const lp::Material &mm = scene.materials.back();
                note_tex(mm.color_tex_idx); note_tex(mm.emission_tex_idx); note_tex(mm.roughness_tex_idx); note_tex(mm.scattering_tex_idx); note_tex(mm.normal_tex_idx);
            }
        }
        else if (section.first == "shapes")
        {
            for (const Json &s : items.arr)
            {
                const Json *uri = s.find("uri");
                if (!uri || uri->str.empty()) continue;
                std::string lower = uri->str;
                for (char &c : lower) c = (char)std::tolower((unsigned char)c);
                if (lower.size() < 4 || lower.compare(lower.size() - 4, 4, ".ply") != 0) throw LoadError("unsupported shape format: " + uri->str);
                load_mesh_ply(find_asset(uri->str, dirs), scene);
            }
        }
        else if (section.first == "instances")
        {
            for (const Json &it : items.arr)
            {
                lp::Mat3x4 transform = mul3x4(conversion, lp::mat3x4_identity());
                uint32_t mesh_idx = 0, mat_idx = 0;
                for (const auto &kv : it.obj)
                {
                    if (kv.first == "frame") transform = mul3x4(conversion, parse_mat3x4(kv.second));
                    else if (kv.first == "material") mat_idx = idx(kv.second);
                    else if (kv.first == "shape") mesh_idx = idx(kv.second);
                }
                scene.instances.push_back(instance_from_transform(transform, mesh_idx, mat_idx));
            }
        }
    }

    const size_t n_tex = std::max<size_t>(tex_paths.size(), tex_referenced);
    tex_paths.resize(n_tex);
    std::vector<std::vector<lp::Vec4>> tex_f32;
    for (const std::string &p : tex_paths)
    {
        if (p.empty()) throw LoadError("texture referenced but not declared");
        LoadedTexture t = load_texture(find_asset(p, dirs));
        out.textures.push_back(std::move(t.texture));
        tex_f32.push_back(std::move(t.f32));
    }
    for (const lp::Environment &env : scene.environments)
    {
        lp::EnvMapInfo info;
        if (env.emission_tex_idx == lp::SENTINEL_IDX) { info.data = {lp::Vec4{1, 1, 1, 1}}; info.width = info.height = 1; }   // loader.rs:728-737
        else
        {
            info.data = tex_f32[env.emission_tex_idx];
            info.width = out.textures[env.emission_tex_idx].width; info.height = out.textures[env.emission_tex_idx].height;
        }
        out.envs_info.push_back(std::move(info));
    }
    lp::validate_scene(scene, (uint32_t)out.textures.size(), (uint32_t)out.textures.size());
    return out;
}

// lpl::load_scene_yoctogl_v24 (loader.rs:331) -> (Scene, cameras)
inline std::pair<lp::Scene, std::vector<SceneCamera>> load_scene_yoctogl_v24(const std::string &path, const lp::Device &d, bool build_both_bvhs = true,
                                                                             const std::vector<std::string> &asset_dirs = {})
{
    LoadedSceneCPU cpu = load_scene_cpu_yoctogl_v24(path, asset_dirs);
    return {lp::build_accel_structures_and_upload(d, cpu.scene, cpu.textures, cpu.envs_info, build_both_bvhs), std::move(cpu.cameras)};
}

}  // namespace lpl
