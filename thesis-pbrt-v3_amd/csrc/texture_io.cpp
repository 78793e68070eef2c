// hprt host side — image textures: file readers (ReadImage, core/imageio.cpp:60-79 for .tga / .png / .pfm)
// and the MIPMap constructor (core/mipmap.h:113-201) with ImageTexture::GetTexture's texel conversion
// (textures/imagemap.cpp:52-97).  Everything here runs once per texture on the host, with the host libm the
// reference uses (pow in the inverse gamma, sin in the Lanczos weights, exp in the EWA table); the kernels and
// the oracle only look the finished pyramid up.
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include "scene_model.h"

namespace hprt {

namespace {

struct FileBytes {
    std::vector<uint8_t> d;
    bool load(const std::string &path) {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) return false;
        fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
        if (n < 0) { fclose(fp); return false; }
        d.resize((size_t)n);
        bool ok = n == 0 || fread(d.data(), 1, (size_t)n, fp) == (size_t)n;
        fclose(fp);
        return ok;
    }
};

bool hasExt(const std::string &name, const char *ext) {
    size_t n = strlen(ext);
    if (name.size() < n) return false;
    for (size_t i = 0; i < n; ++i) if (tolower((unsigned char)name[name.size() - n + i]) != ext[i]) return false;
    return true;
}

// ReadImageTGA (core/imageio.cpp:216-256 over ext/targa.c): rows come out top to bottom, BGR(A) -> RGB, byte / 255.f
bool readTGA(const std::vector<uint8_t> &f, int *w, int *h, std::vector<float> *rgb, std::string *err) {
    if (f.size() < 18) { *err = "truncated TGA header"; return false; }
    const int idLen = f[0], cmapType = f[1], type = f[2];
    const int cmapLen = f[5] | (f[6] << 8), cmapBits = f[7];
    const int W = f[12] | (f[13] << 8), H = f[14] | (f[15] << 8), bpp = f[16], desc = f[17];
    const bool rle = type == 10 || type == 11, mono = type == 3 || type == 11;
    if (!(type == 2 || type == 3 || type == 10 || type == 11) || cmapType != 0) { *err = "TGA type not supported (true-colour or grey, raw or RLE)"; return false; }
    if (!((mono && bpp == 8) || (!mono && (bpp == 24 || bpp == 32))) || W <= 0 || H <= 0) { *err = "TGA pixel depth not supported"; return false; }
    const int bytes = bpp / 8;
    size_t pos = 18 + (size_t)idLen + (size_t)cmapLen * ((cmapBits + 7) / 8);
    std::vector<uint8_t> px((size_t)W * H * bytes);
    if (!rle) {
        if (pos + px.size() > f.size()) { *err = "truncated TGA data"; return false; }
        memcpy(px.data(), &f[pos], px.size());
    } else {
        size_t o = 0;
        while (o < px.size()) {
            if (pos >= f.size()) { *err = "truncated TGA RLE data"; return false; }
            const int c = f[pos++], n = (c & 127) + 1;
            if (c & 128) {
                if (pos + bytes > f.size()) { *err = "truncated TGA RLE data"; return false; }
                for (int i = 0; i < n && o < px.size(); ++i, o += bytes) memcpy(&px[o], &f[pos], bytes);
                pos += bytes;
            } else {
                const size_t nb = (size_t)n * bytes;
                if (pos + nb > f.size() || o + nb > px.size()) { *err = "truncated TGA RLE data"; return false; }
                memcpy(&px[o], &f[pos], nb); pos += nb; o += nb;
            }
        }
    }
    const bool rightToLeft = (desc & 0x10) != 0, topToBottom = (desc & 0x20) != 0;
    *w = W; *h = H; rgb->resize(3 * (size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int sx = rightToLeft ? W - 1 - x : x, sy = topToBottom ? y : H - 1 - y;     // tga_flip_horiz / tga_flip_vert
            const uint8_t *src = &px[((size_t)sy * W + sx) * bytes];
            float *dst = &(*rgb)[3 * ((size_t)y * W + x)];
            if (mono) dst[0] = dst[1] = dst[2] = *src / 255.f;
            else { dst[2] = src[0] / 255.f; dst[1] = src[1] / 255.f; dst[0] = src[2] / 255.f; }
        }
    return true;
}

// ReadImagePFM (core/imageio.cpp:350-431): rows bottom to top in the file, optional byte swap, |scale|
bool readPFM(const std::vector<uint8_t> &f, int *w, int *h, std::vector<float> *rgb, std::string *err) {
    size_t pos = 0;
    auto word = [&](std::string *out) {
        out->clear();
        while (pos < f.size() && !isspace(f[pos])) out->push_back((char)f[pos++]);
        if (pos >= f.size()) return false;
        ++pos;                                   // exactly one whitespace character ends a word (readWord)
        return !out->empty();
    };
    std::string s;
    if (!word(&s) || (s != "Pf" && s != "PF")) { *err = "not a PFM file"; return false; }
    const int nc = s == "PF" ? 3 : 1;
    if (!word(&s)) { *err = "bad PFM header"; return false; }
    const int W = atoi(s.c_str());
    if (!word(&s)) { *err = "bad PFM header"; return false; }
    const int H = atoi(s.c_str());
    if (!word(&s)) { *err = "bad PFM header"; return false; }
    float scale = 0.f; sscanf(s.c_str(), "%f", &scale);
    if (W <= 0 || H <= 0 || pos + 4ull * nc * W * H > f.size()) { *err = "truncated PFM data"; return false; }
    std::vector<float> data((size_t)nc * W * H);
    for (int y = H - 1; y >= 0; --y) { memcpy(&data[(size_t)y * nc * W], &f[pos], 4ull * nc * W); pos += 4ull * nc * W; }
    if (!(scale < 0.f))                          // big-endian file on a little-endian host
        for (float &v : data) { uint8_t b[4]; memcpy(b, &v, 4); std::swap(b[0], b[3]); std::swap(b[1], b[2]); memcpy(&v, b, 4); }
    if (std::abs(scale) != 1.f) for (float &v : data) v *= std::abs(scale);
    *w = W; *h = H; rgb->resize(3 * (size_t)W * H);
    for (size_t i = 0; i < (size_t)W * H; ++i)
        for (int c = 0; c < 3; ++c) (*rgb)[3 * i + c] = nc == 1 ? data[i] : data[3 * i + c];
    return true;
}

// ReadImagePNG (core/imageio.cpp:258-288 over lodepng_decode24): 8-bit RGB from any non-interlaced 8-bit PNG
bool readPNG(const std::vector<uint8_t> &f, int *w, int *h, std::vector<float> *rgb, std::string *err) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) { *err = "not a PNG file"; return false; }
    size_t pos = 8;
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    auto be32 = [&](size_t p) { return ((uint32_t)f[p] << 24) | ((uint32_t)f[p + 1] << 16) | ((uint32_t)f[p + 2] << 8) | f[p + 3]; };
    while (pos + 12 <= f.size()) {
        const uint32_t len = be32(pos);
        const std::string type((const char *)&f[pos + 4], 4);
        if (pos + 12 + (size_t)len > f.size()) { *err = "truncated PNG chunk"; return false; }
        const uint8_t *d = &f[pos + 8];
        if (type == "IHDR" && len >= 13) { W = be32(pos + 8); H = be32(pos + 12); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        pos += 12 + (size_t)len;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (W == 0 || H == 0 || W > 32768 || H > 32768 || depth != 8 || channels == 0 || interlace != 0) { *err = "PNG variant not supported (8-bit, non-interlaced only)"; return false; }
    const size_t stride = (size_t)W * channels;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf outLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outLen, idat.data(), (uLong)idat.size()) != Z_OK || outLen != raw.size()) { *err = "PNG data does not inflate"; return false; }
    std::vector<uint8_t> img(stride * H);
    for (uint32_t y = 0; y < H; ++y) {            // PNG filters, RFC 2083 section 6
        const uint8_t ft = raw[(stride + 1) * y], *in = &raw[(stride + 1) * y + 1];
        uint8_t *out = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)channels ? out[i - channels] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)channels) ? up[i - channels] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) / 2;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) { *err = "bad PNG filter"; return false; }
            out[i] = (uint8_t)(in[i] + pred);
        }
    }
    *w = (int)W; *h = (int)H; rgb->resize(3 * (size_t)W * H);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        uint8_t c[3];
        const uint8_t *p = &img[i * channels];
        if (ctype == 0 || ctype == 4) c[0] = c[1] = c[2] = p[0];
        else if (ctype == 3) { if (3u * p[0] + 2 >= plte.size()) { *err = "PNG palette index out of range"; return false; } memcpy(c, &plte[3 * p[0]], 3); }
        else memcpy(c, p, 3);
        for (int k = 0; k < 3; ++k) (*rgb)[3 * i + k] = c[k] / 255.f;
    }
    return true;
}

inline int modI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }     // Mod, core/pbrt.h
inline int clampI(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline bool isPow2(int v) { return v && !(v & (v - 1)); }
inline int roundUpPow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline float lanczos(float x, float tau = 2.f) {          // core/texture.cpp:254-262
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= 3.14159265358979323846f;
    float s = std::sin(x * tau) / (x * tau);
    float l = std::sin(x) / x;
    return s * l;
}
struct ResampleWeight { int firstTexel; float weight[4]; };
std::vector<ResampleWeight> resampleWeights(int oldRes, int newRes) {      // core/mipmap.h:76-95
    std::vector<ResampleWeight> wt((size_t)newRes);
    const float filterwidth = 2.f;
    for (int i = 0; i < newRes; ++i) {
        float center = (i + .5f) * oldRes / newRes;
        wt[i].firstTexel = (int)std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; ++j) {
            float pos = wt[i].firstTexel + j + .5f;
            wt[i].weight[j] = lanczos((pos - center) / filterwidth);
        }
        float invSumWts = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
    }
    return wt;
}

}  // namespace

bool ReadImageFile(const std::string &path, int *w, int *h, std::vector<float> *rgb, std::string *err) {
    FileBytes f;
    if (!f.load(path)) { *err = "cannot read image file \"" + path + "\""; return false; }
    bool ok;
    if (hasExt(path, ".tga")) ok = readTGA(f.d, w, h, rgb, err);
    else if (hasExt(path, ".png")) ok = readPNG(f.d, w, h, rgb, err);
    else if (hasExt(path, ".pfm")) ok = readPFM(f.d, w, h, rgb, err);
    else { *err = "image format of \"" + path + "\" is not supported (.tga, .png, .pfm)"; return false; }
    if (!ok) *err = path + ": " + *err;
    return ok;
}

// Texel(level, s, t) of a finished level (core/mipmap.h:203-225)
static const float *texelOf(const MipLevel &l, int wrap, int s, int t) {
    static const float black[3] = {0.f, 0.f, 0.f};
    if (wrap == kWrapRepeat) { s = modI(s, l.w); t = modI(t, l.h); }
    else if (wrap == kWrapClamp) { s = clampI(s, 0, l.w - 1); t = clampI(t, 0, l.h - 1); }
    else if (s < 0 || s >= l.w || t < 0 || t >= l.h) return black;
    return &l.rgb[3 * ((size_t)t * l.w + s)];
}

// ImageTexture::GetTexture + MIPMap::MIPMap.  `rgb` is what ReadImage returned (top row first).
void BuildMipMap(int w, int h, const std::vector<float> &rgbIn, float scale, bool gamma, TextureDesc *tex, bool flipY) {
    std::vector<float> img(rgbIn);
    // "Flip image in y; texture coordinate space has (0,0) at the lower left corner" (ImageTexture::GetTexture, textures/imagemap.cpp:
    // 54-60; an InfiniteAreaLight hands its texels to MIPMap as ReadImage returned them: flipY = false)
    if (flipY)
    for (int y = 0; y < h / 2; ++y)
        for (int x = 0; x < 3 * w; ++x) std::swap(img[(size_t)y * 3 * w + x], img[(size_t)(h - 1 - y) * 3 * w + x]);
    // convertIn: scale * (gamma ? InverseGammaCorrect(v) : v), core/pbrt.h:298-301
    for (float &v : img) {
        float c = v;
        if (gamma) c = c <= 0.04045f ? c * 1.f / 12.92f : std::pow((c + 0.055f) * 1.f / 1.055f, (float)2.4f);
        v = scale * c;
    }
    const int wrap = tex->wrap;
    int res[2] = {w, h};
    if (!isPow2(w) || !isPow2(h)) {
        const int p2[2] = {roundUpPow2(w), roundUpPow2(h)};
        std::vector<float> out(3 * (size_t)p2[0] * p2[1], 0.f);
        const std::vector<ResampleWeight> sW = resampleWeights(w, p2[0]);
        for (int t = 0; t < h; ++t)
            for (int s = 0; s < p2[0]; ++s) {
                float *dst = &out[3 * ((size_t)t * p2[0] + s)];
                dst[0] = dst[1] = dst[2] = 0.f;
                for (int j = 0; j < 4; ++j) {
                    int origS = sW[s].firstTexel + j;
                    if (wrap == kWrapRepeat) origS = modI(origS, w);
                    else if (wrap == kWrapClamp) origS = clampI(origS, 0, w - 1);
                    if (origS >= 0 && origS < w)
                        for (int c = 0; c < 3; ++c) dst[c] += sW[s].weight[j] * img[3 * ((size_t)t * w + origS) + c];
                }
            }
        const std::vector<ResampleWeight> tW = resampleWeights(h, p2[1]);
        std::vector<float> work(3 * (size_t)p2[1]);
        for (int s = 0; s < p2[0]; ++s) {
            for (int t = 0; t < p2[1]; ++t) {
                float *wd = &work[3 * (size_t)t];
                wd[0] = wd[1] = wd[2] = 0.f;
                for (int j = 0; j < 4; ++j) {
                    int offset = tW[t].firstTexel + j;
                    if (wrap == kWrapRepeat) offset = modI(offset, h);
                    else if (wrap == kWrapClamp) offset = clampI(offset, 0, h - 1);
                    if (offset >= 0 && offset < h)
                        for (int c = 0; c < 3; ++c) wd[c] += tW[t].weight[j] * out[3 * ((size_t)offset * p2[0] + s) + c];
                }
            }
            for (int t = 0; t < p2[1]; ++t)
                for (int c = 0; c < 3; ++c) { float v = work[3 * (size_t)t + c]; out[3 * ((size_t)t * p2[0] + s) + c] = v < 0.f ? 0.f : v; }   // clamp(v, 0, Infinity)
        }
        img.swap(out); res[0] = p2[0]; res[1] = p2[1];
    }
    int nLevels = 1, m = std::max(res[0], res[1]);
    while ((1 << nLevels) <= m) ++nLevels;                    // 1 + Log2Int(max)
    tex->levels.clear(); tex->levels.resize((size_t)nLevels);
    tex->levels[0].w = res[0]; tex->levels[0].h = res[1]; tex->levels[0].rgb.swap(img);
    for (int i = 1; i < nLevels; ++i) {
        const MipLevel &prev = tex->levels[(size_t)i - 1];
        MipLevel &cur = tex->levels[(size_t)i];
        cur.w = std::max(1, prev.w / 2); cur.h = std::max(1, prev.h / 2);
        cur.rgb.resize(3 * (size_t)cur.w * cur.h);
        for (int t = 0; t < cur.h; ++t)
            for (int s = 0; s < cur.w; ++s) {
                const float *a = texelOf(prev, wrap, 2 * s, 2 * t), *b = texelOf(prev, wrap, 2 * s + 1, 2 * t),
                            *c = texelOf(prev, wrap, 2 * s, 2 * t + 1), *d = texelOf(prev, wrap, 2 * s + 1, 2 * t + 1);
                for (int k = 0; k < 3; ++k) cur.rgb[3 * ((size_t)t * cur.w + s) + k] = .25f * (a[k] + b[k] + c[k] + d[k]);
            }
    }
    for (int i = 0; i < 128; ++i) {                           // EWA weights, core/mipmap.h:193-199
        float alpha = 2;
        float r2 = float(i) / float(128 - 1);
        tex->weightLut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
    }
}

void KeepTextureSource(int w, int h, const std::vector<float> &rgb, float scale, bool gamma, bool flipY, TextureDesc *tex) {
    tex->srcW = w; tex->srcH = h; tex->srcScale = scale; tex->srcGamma = gamma ? 1 : 0; tex->srcFlipY = flipY ? 1 : 0;
    tex->src8.clear(); tex->srcF.clear();
    bool bytes = true;
    std::vector<uint8_t> b(rgb.size());
    for (size_t i = 0; i < rgb.size() && bytes; ++i) {
        const float k = std::nearbyint(rgb[i] * 255.f);
        if (!(k >= 0.f && k <= 255.f) || (float)(int)k / 255.f != rgb[i]) bytes = false;      // (the TGA / PNG readers form byte / 255.f)
        else b[i] = (uint8_t)(int)k;
    }
    if (bytes) tex->src8 = std::move(b); else tex->srcF = rgb;
}
void RebuildFromSource(TextureDesc *tex) {
    std::vector<float> rgb;
    if (!tex->src8.empty()) { rgb.resize(tex->src8.size()); for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (float)tex->src8[i] / 255.f; }
    else rgb = tex->srcF;
    tex->levels.clear();
    BuildMipMap(tex->srcW, tex->srcH, rgb, tex->srcScale, tex->srcGamma != 0, tex, tex->srcFlipY != 0);
}

}  // namespace hprt
