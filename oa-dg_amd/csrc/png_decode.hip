// PNG file -> BGR uint8 [H][W][3] straight into the caller's (pinned) buffer, on the HOST, without the interpreter lock.
//
// Replaces, for the input path of tools/train.py on real files:
//   mmdet/datasets/pipelines/loading.py:33-78  LoadImageFromFile (mmcv.imfrombytes -> cv2.imdecode, colour, BGR)
// A 1024 x 2048 Cityscapes-sized PNG is 30 - 80 ms of inflate + unfilter on one core and a training step consumes four of
// them every ~27 ms: the decode has to run on a dozen threads at once.  PIL decodes in 64 KB pieces and takes the
// interpreter lock back between them (and the caller then copies the image twice: np.asarray + the channel flip); here one
// ctypes call (lock released for its whole duration) inflates the IDAT stream with zlib, undoes the five PNG row filters
// and writes the pixels in OpenCV's channel order at their final place.  PNG is lossless: the bytes equal cv2's / PIL's.
// Covers what such datasets hold: 8-bit grey / RGB / RGBA, non-interlaced; anything else returns OADG_EUNSUPPORTED and the
// caller falls back to PIL.  Host code only (compiled by hipcc with the rest of the library, links zlib).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "common.h"
#include "oadg_hip.h"

namespace {

inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// one scanline in place: cur = filtered bytes (n), prev = the reconstructed line above (or zeros)
void unfilter(int type, unsigned char* cur, const unsigned char* prev, long n, int bpp) {
    switch (type) {
        case 0: break;
        case 1: for (long i = bpp; i < n; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - bpp]); break;
        case 2: for (long i = 0; i < n; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]); break;
        case 3:
            for (long i = 0; i < bpp; ++i) cur[i] = (unsigned char)(cur[i] + (prev[i] >> 1));
            for (long i = bpp; i < n; ++i) cur[i] = (unsigned char)(cur[i] + ((cur[i - bpp] + prev[i]) >> 1));
            break;
        default:
            for (long i = 0; i < bpp; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]);
            for (long i = bpp; i < n; ++i) cur[i] = (unsigned char)(cur[i] + paeth(cur[i - bpp], prev[i], prev[i - bpp]));
    }
}

}  // namespace

extern "C" int oadg_png_size(const char* path, int* height, int* width) {
    if (!path || !height || !width) return OADG_EARG;
    FILE* f = fopen(path, "rb");
    if (!f) return OADG_EIO;
    unsigned char h[33];
    const size_t got = fread(h, 1, 33, f);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (got < 33 || memcmp(h, sig, 8) != 0 || memcmp(h + 12, "IHDR", 4) != 0) return OADG_EUNSUPPORTED;
    *width = (int)be32(h + 16);
    *height = (int)be32(h + 20);
    return OADG_OK;
}

extern "C" int oadg_png_decode_bgr(const char* path, uint8_t* out, int H, int W) {
    if (!path || !out || H < 1 || W < 1) return OADG_EARG;
    FILE* f = fopen(path, "rb");
    if (!f) return OADG_EIO;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size < 57) { fclose(f); return OADG_EUNSUPPORTED; }
    unsigned char* file = (unsigned char*)malloc((size_t)size);
    if (!file) { fclose(f); return OADG_EIO; }
    const size_t got = fread(file, 1, (size_t)size, f);
    fclose(f);
    int rc = OADG_EUNSUPPORTED;
    unsigned char* raw = nullptr;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    do {
        if ((long)got != size || memcmp(file, sig, 8) != 0 || memcmp(file + 12, "IHDR", 4) != 0) break;
        const long w = be32(file + 16), h = be32(file + 20);
        const int depth = file[24], ctype = file[25], interlace = file[28];
        if (w != W || h != H) { rc = OADG_ESIZE; break; }
        if (depth != 8 || interlace != 0 || !(ctype == 0 || ctype == 2 || ctype == 6)) break;
        const int bpp = ctype == 0 ? 1 : (ctype == 2 ? 3 : 4);
        const long line = w * bpp, need = h * (line + 1);
        raw = (unsigned char*)malloc((size_t)need);
        if (!raw) { rc = OADG_EIO; break; }
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) { rc = OADG_EIO; break; }
        zs.next_out = raw;
        zs.avail_out = (uInt)need;
        long pos = 8;
        int zrc = Z_OK;
        bool end = false;
        while (pos + 12 <= size && !end) {
            const unsigned len = be32(file + pos);
            const unsigned char* type = file + pos + 4;
            if (pos + 12 + (long)len > size) break;
            if (memcmp(type, "IDAT", 4) == 0 && zrc == Z_OK) {
                zs.next_in = file + pos + 8;
                zs.avail_in = len;
                zrc = inflate(&zs, Z_NO_FLUSH);
            } else if (memcmp(type, "IEND", 4) == 0) {
                end = true;
            }
            pos += 12 + (long)len;
        }
        const bool complete = (zrc == Z_STREAM_END || zrc == Z_OK) && zs.total_out == (uLong)need;
        inflateEnd(&zs);
        if (!complete) break;
        // unfilter in place (a zero line above the first), then the pixels in B, G, R order
        unsigned char* zero = (unsigned char*)calloc((size_t)line, 1);
        if (!zero) { rc = OADG_EIO; break; }
        const unsigned char* prev = zero;
        bool ok = true;
        for (long y = 0; y < h && ok; ++y) {
            unsigned char* cur = raw + y * (line + 1) + 1;
            const int ft = cur[-1];
            if (ft > 4) { ok = false; break; }
            unfilter(ft, cur, prev, line, bpp);
            uint8_t* o = out + (size_t)y * W * 3;
            if (bpp == 1) for (long x = 0; x < w; ++x) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = cur[x]; }
            else for (long x = 0; x < w; ++x) { o[3 * x] = cur[bpp * x + 2]; o[3 * x + 1] = cur[bpp * x + 1]; o[3 * x + 2] = cur[bpp * x]; }
            prev = cur;
        }
        free(zero);
        if (ok) rc = OADG_OK;
    } while (false);
    free(raw);
    free(file);
    return rc;
}
