// Datastore image I/O fast path (include/m3d_io.h): Zarr v3 chunk files with Blosc 1 frames
// <-> volumes in HBM.  Host threads run the byte-serial entropy stage (zstd / lz4) block by
// block into pinned staging; the shuffle stage and the chunk <-> volume scatter are HIP kernels.
//
// Blosc 1 frame (c-blosc 1.x, the format tensorstore's `blosc` codec writes):
//   [0] format version  [1] codec format version  [2] flags  [3] typesize
//   [4:8] nbytes  [8:12] blocksize  [12:16] cbytes            (little endian)
//   flags: 0x1 byte shuffle, 0x2 memcpyed, 0x4 bit shuffle, 0x10 blocks not split, bits 5-7 codec
//   then int32 bstarts[nblocks] (offsets from the frame start), then per block and per split
//   stream: int32 csize + csize bytes (csize == stream length: stored verbatim).
//   A block is split into `typesize` streams when the 0x10 flag is clear, typesize <= 16,
//   blocksize / typesize >= 128 and it is not the short last block.
//   Bit shuffle (c-blosc 1 rule, checked against libblosc 1.21 frames in tests/golden/blosc_frames.npz):
//   a block whose element count is a multiple of eight is stored as (8 * typesize) rows, row r holding
//   bit r of every element, eight elements per byte, element i at bit i % 8, and bytes beyond the
//   last whole element follow verbatim; any other block is stored unshuffled.
#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/m3d_io.h"
#include "m3d_common.h"

namespace {

using clk = std::chrono::steady_clock;
inline double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// ------------------------------------------------------------------ system codecs (loaded on first use)
struct Codecs {
    size_t (*zstd_compress)(void *, size_t, const void *, size_t, int) = nullptr;
    size_t (*zstd_decompress)(void *, size_t, const void *, size_t) = nullptr;
    size_t (*zstd_bound)(size_t) = nullptr;
    unsigned (*zstd_is_error)(size_t) = nullptr;
    int (*lz4_decompress)(const char *, char *, int, int) = nullptr;
    int (*lz4_compress)(const char *, char *, int, int, int) = nullptr;
    int (*lz4_bound)(int) = nullptr;
    bool have_zstd = false, have_lz4 = false;
};

const Codecs &codecs()
{
    static const Codecs c = [] {
        Codecs k;
        if (void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL)) {
            k.zstd_compress = (decltype(k.zstd_compress))dlsym(h, "ZSTD_compress");
            k.zstd_decompress = (decltype(k.zstd_decompress))dlsym(h, "ZSTD_decompress");
            k.zstd_bound = (decltype(k.zstd_bound))dlsym(h, "ZSTD_compressBound");
            k.zstd_is_error = (decltype(k.zstd_is_error))dlsym(h, "ZSTD_isError");
            k.have_zstd = k.zstd_compress && k.zstd_decompress && k.zstd_bound && k.zstd_is_error;
        }
        if (void *h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL)) {
            k.lz4_decompress = (decltype(k.lz4_decompress))dlsym(h, "LZ4_decompress_safe");
            k.lz4_compress = (decltype(k.lz4_compress))dlsym(h, "LZ4_compress_fast");
            k.lz4_bound = (decltype(k.lz4_bound))dlsym(h, "LZ4_compressBound");
            k.have_lz4 = k.lz4_decompress && k.lz4_compress && k.lz4_bound;
        }
        return k;
    }();
    return c;
}

enum { kLZ4 = 1, kZSTD = 4 };
constexpr uint32_t kHeader = 16;
constexpr size_t kDefaultBlock = 256 * 1024;

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void put32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8), p[2] = (uint8_t)(v >> 16), p[3] = (uint8_t)(v >> 24);
}

int parse_header(const uint8_t *f, size_t n, m3d_blosc_header *h)
{
    M3D_REQUIRE(f && h, "null frame");
    M3D_REQUIRE(n >= kHeader, "blosc frame shorter than its header");
    const uint8_t flags = f[2];
    h->typesize = f[3];
    h->nbytes = le32(f + 4);
    h->blocksize = le32(f + 8);
    h->cbytes = le32(f + 12);
    h->shuffle = (flags & 0x4) ? 2 : ((flags & 0x1) ? 1 : 0);
    h->memcpyed = (flags & 0x2) ? 1 : 0;
    h->split = (flags & 0x10) ? 0 : 1;
    h->codec = flags >> 5;
    M3D_REQUIRE(f[0] == 2, "unsupported blosc format version (expected 2)");
    M3D_REQUIRE(h->typesize >= 1, "blosc typesize 0");
    M3D_REQUIRE(h->cbytes <= n, "blosc frame truncated");
    M3D_REQUIRE(h->nbytes == 0 || h->blocksize > 0, "blosc blocksize 0");
    h->nblocks = h->nbytes ? (int32_t)((h->nbytes + h->blocksize - 1) / h->blocksize) : 0;
    if (h->memcpyed)
        M3D_REQUIRE((size_t)h->nbytes + kHeader <= n, "memcpyed blosc frame truncated");
    else
        M3D_REQUIRE((size_t)kHeader + 4u * (size_t)h->nblocks <= h->cbytes, "blosc block index truncated");
    return M3D_OK;
}

inline uint32_t block_bytes(const m3d_blosc_header &h, int b)
{
    const uint32_t rem = h.nbytes % h.blocksize;
    return (b == h.nblocks - 1 && rem) ? rem : h.blocksize;
}

// effective shuffle of one block (c-blosc 1: byte shuffle needs typesize > 1; bit shuffle needs at least one
// element and an element count that is a multiple of eight)
__host__ __device__ inline int effective_shuffle(int shuffle, int T, uint32_t bsize)
{
    if (shuffle == 1) return T > 1 ? 1 : 0;
    if (shuffle == 2) return (bsize >= (uint32_t)T && (bsize / (uint32_t)T) % 8 == 0) ? 2 : 0;
    return 0;
}
inline int block_shuffle(const m3d_blosc_header &h, uint32_t bsize) { return effective_shuffle(h.shuffle, h.typesize, bsize); }

// Entropy stage of one block: `out` receives the block still shuffled.
int decode_block_streams(const uint8_t *f, const m3d_blosc_header &h, int b, uint8_t *out)
{
    const Codecs &c = codecs();
    const uint32_t bsize = block_bytes(h, b);
    const bool leftover = bsize != h.blocksize;
    const int nsplits =
        (h.split && h.typesize <= 16 && h.blocksize / (uint32_t)h.typesize >= 128 && !leftover) ? h.typesize : 1;
    const uint32_t neblock = bsize / (uint32_t)nsplits;
    size_t pos = le32(f + kHeader + 4 * (size_t)b);
    for (int j = 0; j < nsplits; ++j) {
        M3D_REQUIRE(pos + 4 <= h.cbytes, "blosc stream header out of bounds");
        const uint32_t cs = le32(f + pos);
        pos += 4;
        M3D_REQUIRE(pos + cs <= h.cbytes, "blosc stream out of bounds");
        uint8_t *dst = out + (size_t)j * neblock;
        if (cs == neblock) {
            memcpy(dst, f + pos, neblock);
        } else if (h.codec == kZSTD) {
            if (!c.have_zstd) {
                m3d::set_err("unsupported", "libzstd.so.1 not found");
                return M3D_ERR_UNSUPPORTED;
            }
            const size_t got = c.zstd_decompress(dst, neblock, f + pos, cs);
            M3D_REQUIRE(!c.zstd_is_error(got) && got == neblock, "zstd stream does not decode to the block size");
        } else if (h.codec == kLZ4) {
            if (!c.have_lz4) {
                m3d::set_err("unsupported", "liblz4.so.1 not found");
                return M3D_ERR_UNSUPPORTED;
            }
            const int got = c.lz4_decompress((const char *)(f + pos), (char *)dst, (int)cs, (int)neblock);
            M3D_REQUIRE(got == (int)neblock, "lz4 stream does not decode to the block size");
        } else {
            m3d::set_err("unsupported", "blosc codec other than zstd / lz4");
            return M3D_ERR_UNSUPPORTED;
        }
        pos += cs;
    }
    return M3D_OK;
}

// ------------------------------------------------------------------ host shuffle stage
inline uint64_t transpose8x8(uint64_t x)
{
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull, x = x ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull, x = x ^ t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull, x = x ^ t ^ (t << 28);
    return x;
}

void host_unshuffle(const uint8_t *src, uint8_t *dst, uint32_t bsize, int T, int mode)
{
    if (mode == 0) {
        memcpy(dst, src, bsize);
        return;
    }
    const uint32_t ne = bsize / (uint32_t)T;
    if (mode == 1) {
        for (int j = 0; j < T; ++j)
            for (uint32_t i = 0; i < ne; ++i) dst[(size_t)i * T + j] = src[(size_t)j * ne + i];
        memcpy(dst + (size_t)ne * T, src + (size_t)ne * T, bsize - ne * T);
        return;
    }
    const uint32_t n8 = ne - ne % 8, nq = n8 / 8;
    for (int j = 0; j < T; ++j)
        for (uint32_t q = 0; q < nq; ++q) {
            uint64_t x = 0;  // byte k = row (8j + k) at column q: bit m of it belongs to element 8q + m
            for (int k = 0; k < 8; ++k) x |= (uint64_t)src[(size_t)(8 * j + k) * nq + q] << (8 * k);
            x = transpose8x8(x);  // byte m = byte j of element 8q + m
            for (int m = 0; m < 8; ++m) dst[((size_t)q * 8 + m) * T + j] = (uint8_t)(x >> (8 * m));
        }
    memcpy(dst + (size_t)n8 * T, src + (size_t)n8 * T, bsize - n8 * T);
}

void host_shuffle(const uint8_t *src, uint8_t *dst, uint32_t bsize, int T, int mode)
{
    if (mode == 0) {
        memcpy(dst, src, bsize);
        return;
    }
    const uint32_t ne = bsize / (uint32_t)T;
    if (mode == 1) {
        for (int j = 0; j < T; ++j)
            for (uint32_t i = 0; i < ne; ++i) dst[(size_t)j * ne + i] = src[(size_t)i * T + j];
        memcpy(dst + (size_t)ne * T, src + (size_t)ne * T, bsize - ne * T);
        return;
    }
    const uint32_t n8 = ne - ne % 8, nq = n8 / 8;
    for (int j = 0; j < T; ++j)
        for (uint32_t q = 0; q < nq; ++q) {
            uint64_t x = 0;
            for (int m = 0; m < 8; ++m) x |= (uint64_t)src[((size_t)q * 8 + m) * T + j] << (8 * m);
            x = transpose8x8(x);
            for (int k = 0; k < 8; ++k) dst[(size_t)(8 * j + k) * nq + q] = (uint8_t)(x >> (8 * k));
        }
    memcpy(dst + (size_t)n8 * T, src + (size_t)n8 * T, bsize - n8 * T);
}

// ------------------------------------------------------------------ entropy stage, encode
// One frame from data that is already shuffled block by block (`shuffled`, nbytes long).
int encode_frame(const uint8_t *shuffled, size_t nbytes, int T, int clevel, int shuffle, int codec, size_t blocksize,
                 uint8_t *dst, size_t cap, size_t *out_bytes, std::vector<uint8_t> &scratch)
{
    const Codecs &c = codecs();
    M3D_REQUIRE(nbytes <= 0x7fffffffu - kHeader, "blosc buffers are limited to 2 GiB");
    M3D_REQUIRE(codec == kZSTD || codec == kLZ4, "codec must be 1 (lz4) or 4 (zstd)");
    if ((codec == kZSTD && !c.have_zstd) || (codec == kLZ4 && !c.have_lz4)) {
        m3d::set_err("unsupported", codec == kZSTD ? "libzstd.so.1 not found" : "liblz4.so.1 not found");
        return M3D_ERR_UNSUPPORTED;
    }
    const uint32_t nblocks = nbytes ? (uint32_t)((nbytes + blocksize - 1) / blocksize) : 0;
    M3D_REQUIRE(cap >= m3d_blosc_bound(nbytes, blocksize), "destination smaller than m3d_blosc_bound");
    dst[0] = 2;
    dst[1] = 1;
    dst[2] = (uint8_t)((shuffle == 2 ? 0x4 : (shuffle == 1 ? 0x1 : 0)) | 0x10 | (codec << 5));
    dst[3] = (uint8_t)T;
    put32(dst + 4, (uint32_t)nbytes);
    put32(dst + 8, (uint32_t)blocksize);
    size_t pos = kHeader + 4 * (size_t)nblocks;
    const size_t bound = codec == kZSTD ? c.zstd_bound(blocksize) : (size_t)c.lz4_bound((int)blocksize);
    if (scratch.size() < bound) scratch.resize(bound);
    for (uint32_t b = 0; b < nblocks; ++b) {
        const size_t off = (size_t)b * blocksize;
        const size_t bsize = off + blocksize <= nbytes ? blocksize : nbytes - off;
        put32(dst + kHeader + 4 * (size_t)b, (uint32_t)pos);
        size_t cs;
        if (codec == kZSTD) {
            cs = c.zstd_compress(scratch.data(), bound, shuffled + off, bsize, clevel < 9 ? 2 * clevel - 1 : 19);
            if (c.zstd_is_error(cs)) cs = bsize;
        } else {
            const int r = c.lz4_compress((const char *)(shuffled + off), (char *)scratch.data(), (int)bsize, (int)bound,
                                         clevel >= 9 ? 1 : 10 - clevel);
            cs = r > 0 ? (size_t)r : bsize;
        }
        if (cs >= bsize) {  // incompressible: stored verbatim (csize == block length marks it)
            put32(dst + pos, (uint32_t)bsize);
            memcpy(dst + pos + 4, shuffled + off, bsize);
            pos += 4 + bsize;
        } else {
            put32(dst + pos, (uint32_t)cs);
            memcpy(dst + pos + 4, scratch.data(), cs);
            pos += 4 + cs;
        }
    }
    put32(dst + 12, (uint32_t)pos);
    *out_bytes = pos;
    return M3D_OK;
}

// ------------------------------------------------------------------ files
bool read_file(const std::string &path, std::vector<uint8_t> &buf, size_t *n, bool *missing)
{
    *missing = false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) {
        *missing = errno == ENOENT;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return false;
    }
    const size_t size = (size_t)st.st_size;
    if (buf.size() < size) buf.resize(size);
    size_t got = 0;
    while (got < size) {
        const ssize_t r = read(fd, buf.data() + got, size - got);
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    *n = got;
    return got == size;
}

bool write_file(const std::string &path, const uint8_t *data, size_t n)
{
    const std::string tmp = path + ".partial";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    size_t put = 0;
    while (put < n) {
        const ssize_t r = write(fd, data + put, n - put);
        if (r <= 0) break;
        put += (size_t)r;
    }
    close(fd);
    return put == n && rename(tmp.c_str(), path.c_str()) == 0;
}

void make_dirs(const std::string &path)
{
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0755);
}

// ------------------------------------------------------------------ chunk grid
struct Grid {
    int64_t shape[3], chunk[3], n[3];
    int T;
    int64_t nchunks, chunk_elems, chunk_bytes;
};

int make_grid(const int64_t *shape, const int64_t *chunk, int T, Grid *g)
{
    M3D_REQUIRE(shape && chunk, "null shape");
    M3D_REQUIRE(T == 1 || T == 2 || T == 4 || T == 8, "typesize must be 1, 2, 4 or 8");
    g->T = T;
    g->chunk_elems = 1;
    g->nchunks = 1;
    for (int a = 0; a < 3; ++a) {
        M3D_REQUIRE(shape[a] > 0 && chunk[a] > 0, "shape and chunk extents must be positive");
        g->shape[a] = shape[a], g->chunk[a] = chunk[a];
        g->n[a] = (shape[a] + chunk[a] - 1) / chunk[a];
        g->chunk_elems *= chunk[a];
        g->nchunks *= g->n[a];
    }
    g->chunk_bytes = g->chunk_elems * T;
    M3D_REQUIRE(g->chunk_bytes <= 0x7fffffff - 64, "chunk larger than a blosc buffer (2 GiB)");
    return M3D_OK;
}

inline void chunk_coords(const Grid &g, int64_t c, int64_t *cz, int64_t *cy, int64_t *cx)
{
    *cx = c % g.n[2];
    *cy = (c / g.n[2]) % g.n[1];
    *cz = c / (g.n[2] * g.n[1]);
}

inline std::string chunk_path(const std::string &dir, int64_t cz, int64_t cy, int64_t cx)
{
    return dir + "/" + std::to_string(cz) + "/" + std::to_string(cy) + "/" + std::to_string(cx);
}

// host scatter / gather of one decoded chunk
void host_scatter(const Grid &g, int64_t c, const uint8_t *chunk, uint8_t *vol)
{
    int64_t cz, cy, cx;
    chunk_coords(g, c, &cz, &cy, &cx);
    const int64_t z0 = cz * g.chunk[0], y0 = cy * g.chunk[1], x0 = cx * g.chunk[2];
    const int64_t nx = std::min(g.chunk[2], g.shape[2] - x0);
    for (int64_t z = 0; z < g.chunk[0] && z0 + z < g.shape[0]; ++z)
        for (int64_t y = 0; y < g.chunk[1] && y0 + y < g.shape[1]; ++y)
            memcpy(vol + (((z0 + z) * g.shape[1] + (y0 + y)) * g.shape[2] + x0) * g.T,
                   chunk + ((z * g.chunk[1] + y) * g.chunk[2]) * g.T, (size_t)(nx * g.T));
}

void host_gather(const Grid &g, int64_t c, const uint8_t *vol, uint8_t *chunk)
{
    int64_t cz, cy, cx;
    chunk_coords(g, c, &cz, &cy, &cx);
    const int64_t z0 = cz * g.chunk[0], y0 = cy * g.chunk[1], x0 = cx * g.chunk[2];
    const int64_t nx = std::min(g.chunk[2], g.shape[2] - x0);
    memset(chunk, 0, (size_t)g.chunk_bytes);
    for (int64_t z = 0; z < g.chunk[0] && z0 + z < g.shape[0]; ++z)
        for (int64_t y = 0; y < g.chunk[1] && y0 + y < g.shape[1]; ++y)
            memcpy(chunk + ((z * g.chunk[1] + y) * g.chunk[2]) * g.T,
                   vol + (((z0 + z) * g.shape[1] + (y0 + y)) * g.shape[2] + x0) * g.T, (size_t)(nx * g.T));
}

// ------------------------------------------------------------------ device shuffle stage
struct ChunkJob {
    int64_t vz, vy, vx;  // volume extents
    int cz, cy, cx;      // chunk extents
    int64_t oz, oy, ox;  // chunk origin in the volume
    uint32_t nbytes, blocksize;
    int shuffle;         // frame-level shuffle mode (0 also covers memcpyed frames)
    uint32_t per_block;  // workgroups per Blosc block (set by the launchers)
};

// One thread = eight consecutive elements of one block.  T-byte elements, 8T rows per block.
template <int T>
__global__ __launch_bounds__(256) void k_unshuffle_scatter(const uint8_t *__restrict__ stage, uint8_t *__restrict__ vol,
                                                           ChunkJob j)
{
    // one-dimensional grid: `per_block` workgroups for each Blosc block (frames may hold many small blocks)
    const uint32_t b = blockIdx.x / j.per_block;
    const uint32_t boff = b * j.blocksize;
    const uint32_t bsize = min(j.blocksize, j.nbytes - boff);
    const uint32_t ne = bsize / T, n8 = ne - ne % 8, nq = n8 / 8;
    const uint32_t q = (blockIdx.x % j.per_block) * blockDim.x + threadIdx.x;
    if (q * 8 >= ne) return;
    const int mode = effective_shuffle(j.shuffle, T, bsize);
    const uint8_t *src = stage + boff;
    uint8_t e[8][T];
    const uint32_t cnt = min(8u, ne - q * 8);
    if (mode == 2 && q < nq) {
#pragma unroll
        for (int jb = 0; jb < T; ++jb) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) lo |= (uint32_t)src[(size_t)(8 * jb + k) * nq + q] << (8 * k);
#pragma unroll
            for (int k = 0; k < 4; ++k) hi |= (uint32_t)src[(size_t)(8 * jb + 4 + k) * nq + q] << (8 * k);
            uint64_t x = ((uint64_t)hi << 32) | lo, t;
            t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull, x = x ^ t ^ (t << 7);
            t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull, x = x ^ t ^ (t << 14);
            t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull, x = x ^ t ^ (t << 28);
#pragma unroll
            for (int m = 0; m < 8; ++m) e[m][jb] = (uint8_t)(x >> (8 * m));
        }
    } else if (mode == 1) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int jb = 0; jb < T; ++jb) e[m][jb] = m < (int)cnt ? src[(size_t)jb * ne + q * 8 + m] : 0;
    } else {  // unshuffled, or the verbatim tail of a bit-shuffled block
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int jb = 0; jb < T; ++jb) e[m][jb] = m < (int)cnt ? src[((size_t)q * 8 + m) * T + jb] : 0;
    }
    const int64_t first = (int64_t)(boff / T) + (int64_t)q * 8;  // element index inside the chunk
    const int64_t ix0 = first % j.cx, iy0 = (first / j.cx) % j.cy, iz0 = first / ((int64_t)j.cx * j.cy);
    if (cnt == 8 && ix0 + 8 <= j.cx && j.ox + ix0 + 8 <= j.vx && j.oy + iy0 < j.vy && j.oz + iz0 < j.vz) {
        uint8_t *dst = vol + (((j.oz + iz0) * j.vy + (j.oy + iy0)) * j.vx + (j.ox + ix0)) * T;
        if constexpr (T >= 2) if (((uintptr_t)dst & 15) == 0) {
            uint32_t w[2 * T];
#pragma unroll
            for (int k = 0; k < 2 * T; ++k) {
                uint32_t v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) v |= (uint32_t)e[(4 * k + bb) / T][(4 * k + bb) % T] << (8 * bb);
                w[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 2 * T; k += 4) *(uint4 *)(dst + 4 * k) = make_uint4(w[k], w[k + 1], w[k + 2], w[k + 3]);
            return;
        }
    }
    for (uint32_t m = 0; m < cnt; ++m) {
        const int64_t idx = first + m;
        const int64_t ix = idx % j.cx, iy = (idx / j.cx) % j.cy, iz = idx / ((int64_t)j.cx * j.cy);
        if (iz >= j.cz || j.oz + iz >= j.vz || j.oy + iy >= j.vy || j.ox + ix >= j.vx) continue;
        uint8_t *dst = vol + (((j.oz + iz) * j.vy + (j.oy + iy)) * j.vx + (j.ox + ix)) * T;
#pragma unroll
        for (int jb = 0; jb < T; ++jb) dst[jb] = e[m][jb];
    }
}

template <int T>
__global__ __launch_bounds__(256) void k_gather_shuffle(const uint8_t *__restrict__ vol, uint8_t *__restrict__ stage,
                                                        ChunkJob j)
{
    // one-dimensional grid: `per_block` workgroups for each Blosc block (frames may hold many small blocks)
    const uint32_t b = blockIdx.x / j.per_block;
    const uint32_t boff = b * j.blocksize;
    const uint32_t bsize = min(j.blocksize, j.nbytes - boff);
    const uint32_t ne = bsize / T, n8 = ne - ne % 8, nq = n8 / 8;
    const uint32_t q = (blockIdx.x % j.per_block) * blockDim.x + threadIdx.x;
    if (q * 8 >= ne) return;
    const int mode = effective_shuffle(j.shuffle, T, bsize);
    uint8_t *dst = stage + boff;
    const uint32_t cnt = min(8u, ne - q * 8);
    const int64_t first = (int64_t)(boff / T) + (int64_t)q * 8;
    uint8_t e[8][T];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int64_t idx = first + m;
        const int64_t ix = idx % j.cx, iy = (idx / j.cx) % j.cy, iz = idx / ((int64_t)j.cx * j.cy);
        const bool in = m < (int)cnt && iz < j.cz && j.oz + iz < j.vz && j.oy + iy < j.vy && j.ox + ix < j.vx;
        const uint8_t *s = vol + (((j.oz + iz) * j.vy + (j.oy + iy)) * j.vx + (j.ox + ix)) * T;
#pragma unroll
        for (int jb = 0; jb < T; ++jb) e[m][jb] = in ? s[jb] : 0;  // edge chunks are zero padded
    }
    if (mode == 2 && q < nq) {
#pragma unroll
        for (int jb = 0; jb < T; ++jb) {
            uint64_t x = 0, t;
#pragma unroll
            for (int m = 0; m < 8; ++m) x |= (uint64_t)e[m][jb] << (8 * m);
            t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull, x = x ^ t ^ (t << 7);
            t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull, x = x ^ t ^ (t << 14);
            t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull, x = x ^ t ^ (t << 28);
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[(size_t)(8 * jb + k) * nq + q] = (uint8_t)(x >> (8 * k));
        }
    } else if (mode == 1) {
        for (uint32_t m = 0; m < cnt; ++m)
#pragma unroll
            for (int jb = 0; jb < T; ++jb) dst[(size_t)jb * ne + q * 8 + m] = e[m][jb];
    } else {
        for (uint32_t m = 0; m < cnt; ++m)
#pragma unroll
            for (int jb = 0; jb < T; ++jb) dst[((size_t)q * 8 + m) * T + jb] = e[m][jb];
    }
}

template <int T>
void launch_unshuffle(const uint8_t *stage, uint8_t *vol, ChunkJob j, hipStream_t st)
{
    const uint32_t nblocks = (j.nbytes + j.blocksize - 1) / j.blocksize;
    const uint32_t groups = (j.blocksize / T + 7) / 8;
    j.per_block = (groups + 255) / 256;
    hipLaunchKernelGGL(k_unshuffle_scatter<T>, dim3(j.per_block * nblocks), dim3(256), 0, st, stage, vol, j);
}
template <int T>
void launch_shuffle(const uint8_t *vol, uint8_t *stage, ChunkJob j, hipStream_t st)
{
    const uint32_t nblocks = (j.nbytes + j.blocksize - 1) / j.blocksize;
    const uint32_t groups = (j.blocksize / T + 7) / 8;
    j.per_block = (groups + 255) / 256;
    hipLaunchKernelGGL(k_gather_shuffle<T>, dim3(j.per_block * nblocks), dim3(256), 0, st, vol, stage, j);
}

__global__ void k_zero_box(uint8_t *vol, ChunkJob j, int T)
{
    const int64_t n = (int64_t)j.cz * j.cy * j.cx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ix = i % j.cx, iy = (i / j.cx) % j.cy, iz = i / ((int64_t)j.cx * j.cy);
        if (j.oz + iz >= j.vz || j.oy + iy >= j.vy || j.ox + ix >= j.vx) continue;
        uint8_t *d = vol + (((j.oz + iz) * j.vy + (j.oy + iy)) * j.vx + (j.ox + ix)) * T;
        for (int b = 0; b < T; ++b) d[b] = 0;
    }
}

ChunkJob job_for(const Grid &g, int64_t c)
{
    int64_t cz, cy, cx;
    chunk_coords(g, c, &cz, &cy, &cx);
    ChunkJob j;
    j.vz = g.shape[0], j.vy = g.shape[1], j.vx = g.shape[2];
    j.cz = (int)g.chunk[0], j.cy = (int)g.chunk[1], j.cx = (int)g.chunk[2];
    j.oz = cz * g.chunk[0], j.oy = cy * g.chunk[1], j.ox = cx * g.chunk[2];
    j.nbytes = (uint32_t)g.chunk_bytes;
    j.blocksize = (uint32_t)g.chunk_bytes;
    j.shuffle = 0;
    j.per_block = 1;
    return j;
}

}  // namespace

// ====================================================================== state
struct m3d_io {
    int nthreads = 1;
    uint8_t *pinned = nullptr;
    uint8_t *dev = nullptr;
    size_t pinned_bytes = 0, dev_bytes = 0;
};

namespace {

int ensure_staging(m3d_io *io, size_t bytes, bool device)
{
    if (io->pinned_bytes < bytes) {
        if (io->pinned) (void)hipHostFree(io->pinned);
        io->pinned = nullptr, io->pinned_bytes = 0;
        M3D_HIP_CHECK(hipHostMalloc((void **)&io->pinned, bytes, hipHostMallocDefault));
        io->pinned_bytes = bytes;
    }
    if (device && io->dev_bytes < bytes) {
        if (io->dev) (void)hipFree(io->dev);
        io->dev = nullptr, io->dev_bytes = 0;
        M3D_HIP_CHECK(hipMalloc((void **)&io->dev, bytes));
        io->dev_bytes = bytes;
    }
    return M3D_OK;
}

// state of one chunk as the workers hand it to the issuing thread
struct ChunkState {
    std::atomic<int> ready{0};  // 1 done, -1 failed
    m3d_blosc_header h{};
    bool missing = false;
};

template <typename F>
void run_workers(int nthreads, int64_t nitems, F &&body)
{
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    const int n = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, nitems));
    for (int t = 0; t < n; ++t)
        pool.emplace_back([&, t] {
            for (int64_t i; (i = next.fetch_add(1)) < nitems;) body(i, t);
        });
    for (auto &th : pool) th.join();
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

int m3d_blosc_info(const void *frame, size_t frame_bytes, m3d_blosc_header *out)
{
    return parse_header((const uint8_t *)frame, frame_bytes, out);
}

int m3d_blosc_decode_host(const void *frame, size_t frame_bytes, void *dst, size_t dst_bytes)
{
    m3d_blosc_header h;
    const uint8_t *f = (const uint8_t *)frame;
    if (int rc = parse_header(f, frame_bytes, &h)) return rc;
    M3D_REQUIRE(dst && dst_bytes == h.nbytes, "destination size differs from the frame's nbytes");
    if (h.memcpyed) {
        memcpy(dst, f + kHeader, h.nbytes);
        return M3D_OK;
    }
    std::vector<uint8_t> tmp(h.blocksize);
    for (int b = 0; b < h.nblocks; ++b) {
        const uint32_t bsize = block_bytes(h, b);
        if (int rc = decode_block_streams(f, h, b, tmp.data())) return rc;
        host_unshuffle(tmp.data(), (uint8_t *)dst + (size_t)b * h.blocksize, bsize, h.typesize, block_shuffle(h, bsize));
    }
    return M3D_OK;
}

size_t m3d_blosc_bound(size_t nbytes, size_t blocksize)
{
    if (blocksize == 0) blocksize = kDefaultBlock;
    const size_t nblocks = nbytes ? (nbytes + blocksize - 1) / blocksize : 0;
    return kHeader + 8 * nblocks + nbytes;
}

int m3d_blosc_encode_host(const void *src, size_t nbytes, int typesize, int clevel, int shuffle, int codec,
                          size_t blocksize, void *dst, size_t cap, size_t *out_bytes)
{
    M3D_REQUIRE(src && dst && out_bytes, "null buffer");
    M3D_REQUIRE(typesize >= 1 && typesize <= 255, "typesize out of range");
    M3D_REQUIRE(clevel >= 1 && clevel <= 9, "clevel must be 1..9");
    M3D_REQUIRE(shuffle >= 0 && shuffle <= 2, "shuffle must be 0, 1 or 2");
    if (blocksize == 0) blocksize = kDefaultBlock;
    if (blocksize > nbytes && nbytes) blocksize = nbytes;
    M3D_REQUIRE(blocksize % (size_t)typesize == 0 || blocksize == nbytes, "blocksize must be a multiple of typesize");
    std::vector<uint8_t> shuffled(nbytes), scratch;
    m3d_blosc_header h{};
    h.typesize = typesize, h.shuffle = shuffle;
    for (size_t off = 0; off < nbytes; off += blocksize) {
        const uint32_t bsize = (uint32_t)std::min(blocksize, nbytes - off);
        host_shuffle((const uint8_t *)src + off, shuffled.data() + off, bsize, typesize, block_shuffle(h, bsize));
    }
    return encode_frame(shuffled.data(), nbytes, typesize, clevel, shuffle, codec, blocksize, (uint8_t *)dst, cap,
                        out_bytes, scratch);
}

int m3d_io_create(int nthreads, m3d_io **out)
{
    M3D_REQUIRE(out, "null out");
    M3D_REQUIRE(nthreads >= 1 && nthreads <= 256, "nthreads must be 1..256");
    *out = new m3d_io();
    (*out)->nthreads = nthreads;
    return M3D_OK;
}

void m3d_io_destroy(m3d_io *io)
{
    if (!io) return;
    if (io->pinned) (void)hipHostFree(io->pinned);
    if (io->dev) (void)hipFree(io->dev);
    delete io;
}

static int read_impl(m3d_io *io, const char *chunk_dir, const int64_t *shape, const int64_t *chunk, int T, void *dst,
                     bool device, hipStream_t st, m3d_io_stats *stats)
{
    M3D_REQUIRE(io && chunk_dir && dst, "null argument");
    Grid g;
    if (int rc = make_grid(shape, chunk, T, &g)) return rc;
    const auto t0 = clk::now();
    const std::string dir(chunk_dir);
    std::vector<ChunkState> state((size_t)g.nchunks);
    std::vector<double> busy((size_t)io->nthreads, 0.0);
    std::atomic<int64_t> file_bytes{0};
    std::vector<std::string> errors((size_t)io->nthreads);
    std::vector<int> codes((size_t)io->nthreads, 0);
    uint8_t *pinned = nullptr;
    if (device) {
        if (int rc = ensure_staging(io, (size_t)(g.nchunks * g.chunk_bytes), true)) return rc;
        pinned = io->pinned;
    }

    auto worker = [&](int64_t c, int t) {
        const auto w0 = clk::now();
        thread_local std::vector<uint8_t> file, tmp, full;
        ChunkState &s = state[(size_t)c];
        int64_t cz, cy, cx;
        chunk_coords(g, c, &cz, &cy, &cx);
        size_t n = 0;
        int rc = M3D_OK;
        if (!read_file(chunk_path(dir, cz, cy, cx), file, &n, &s.missing)) {
            if (!s.missing) {
                m3d::set_err("read failed", chunk_path(dir, cz, cy, cx).c_str());
                rc = M3D_ERR_INVALID_ARG;
            }
        } else {
            file_bytes += (int64_t)n;
            rc = parse_header(file.data(), n, &s.h);
            if (!rc && (s.h.nbytes != (uint32_t)g.chunk_bytes || s.h.typesize != T)) {
                m3d::set_err("invalid argument", "chunk frame does not hold chunk_shape x typesize bytes");
                rc = M3D_ERR_INVALID_ARG;
            }
            if (!rc) {
                if (device) {
                    uint8_t *out = pinned + (size_t)c * g.chunk_bytes;  // still shuffled
                    if (s.h.memcpyed)
                        memcpy(out, file.data() + kHeader, s.h.nbytes);
                    else
                        for (int b = 0; b < s.h.nblocks && !rc; ++b)
                            rc = decode_block_streams(file.data(), s.h, b, out + (size_t)b * s.h.blocksize);
                } else {
                    if (full.size() < (size_t)g.chunk_bytes) full.resize((size_t)g.chunk_bytes);
                    if (s.h.memcpyed) {
                        memcpy(full.data(), file.data() + kHeader, s.h.nbytes);
                    } else {
                        if (tmp.size() < s.h.blocksize) tmp.resize(s.h.blocksize);
                        for (int b = 0; b < s.h.nblocks && !rc; ++b) {
                            const uint32_t bsize = block_bytes(s.h, b);
                            rc = decode_block_streams(file.data(), s.h, b, tmp.data());
                            if (!rc)
                                host_unshuffle(tmp.data(), full.data() + (size_t)b * s.h.blocksize, bsize, T,
                                               block_shuffle(s.h, bsize));
                        }
                    }
                    if (!rc) host_scatter(g, c, full.data(), (uint8_t *)dst);
                }
            }
        }
        if (!rc && s.missing && !device) {  // Zarr: an absent chunk reads as the fill value (0)
            if (full.size() < (size_t)g.chunk_bytes) full.resize((size_t)g.chunk_bytes);
            memset(full.data(), 0, (size_t)g.chunk_bytes);
            host_scatter(g, c, full.data(), (uint8_t *)dst);
        }
        if (rc) {
            codes[(size_t)t] = rc;
            errors[(size_t)t] = m3d::err_buf();
        }
        busy[(size_t)t] += since(w0);
        s.ready.store(rc ? -1 : 1, std::memory_order_release);
    };

    int rc = M3D_OK;
    if (!device) {
        run_workers(io->nthreads, g.nchunks, worker);
    } else {
        // the calling thread issues copies and kernels chunk by chunk while the workers decode ahead
        std::thread feeder([&] { run_workers(io->nthreads, g.nchunks, worker); });
        hipError_t herr = hipSuccess;
        for (int64_t c = 0; c < g.nchunks; ++c) {
            ChunkState &s = state[(size_t)c];
            int r;
            while ((r = s.ready.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
            if (r < 0 || herr != hipSuccess) continue;
            ChunkJob j = job_for(g, c);
            if (s.missing) {
                hipLaunchKernelGGL(k_zero_box, dim3(256), dim3(256), 0, st, (uint8_t *)dst, j, T);
                continue;
            }
            uint8_t *d = io->dev + (size_t)c * g.chunk_bytes;
            herr = hipMemcpyAsync(d, pinned + (size_t)c * g.chunk_bytes, (size_t)g.chunk_bytes, hipMemcpyHostToDevice, st);
            if (herr != hipSuccess) continue;
            j.blocksize = s.h.memcpyed ? s.h.nbytes : s.h.blocksize;
            j.shuffle = s.h.memcpyed ? 0 : s.h.shuffle;
            switch (T) {
            case 1: launch_unshuffle<1>(d, (uint8_t *)dst, j, st); break;
            case 2: launch_unshuffle<2>(d, (uint8_t *)dst, j, st); break;
            case 4: launch_unshuffle<4>(d, (uint8_t *)dst, j, st); break;
            default: launch_unshuffle<8>(d, (uint8_t *)dst, j, st); break;
            }
        }
        feeder.join();
        M3D_HIP_CHECK(herr);
        M3D_KERNEL_CHECK();
        M3D_HIP_CHECK(hipStreamSynchronize(st));
    }
    int missing = 0;
    for (auto &s : state) missing += s.missing ? 1 : 0;
    for (int t = 0; t < io->nthreads; ++t)
        if (codes[(size_t)t]) {
            rc = codes[(size_t)t];
            snprintf(m3d::err_buf(), 512, "%s", errors[(size_t)t].c_str());
            break;
        }
    if (stats) {
        stats->seconds_total = since(t0);
        stats->seconds_entropy = 0;
        for (double b : busy) stats->seconds_entropy += b;
        stats->file_bytes = file_bytes.load();
        stats->raw_bytes = g.shape[0] * g.shape[1] * g.shape[2] * T;
        stats->chunks = (int32_t)g.nchunks;
        stats->chunks_missing = missing;
    }
    return rc;
}

int m3d_zarr_read_device(m3d_io *io, const char *chunk_dir, const int64_t shape[3], const int64_t chunk[3], int typesize,
                         void *dst, void *stream, m3d_io_stats *stats)
{
    return read_impl(io, chunk_dir, shape, chunk, typesize, dst, true, (hipStream_t)stream, stats);
}

int m3d_zarr_read_host(m3d_io *io, const char *chunk_dir, const int64_t shape[3], const int64_t chunk[3], int typesize,
                       void *dst, m3d_io_stats *stats)
{
    return read_impl(io, chunk_dir, shape, chunk, typesize, dst, false, nullptr, stats);
}

static int write_impl(m3d_io *io, const char *chunk_dir, const int64_t *shape, const int64_t *chunk, int T,
                      const void *src, int clevel, int shuffle, int codec, bool device, hipStream_t st,
                      m3d_io_stats *stats)
{
    M3D_REQUIRE(io && chunk_dir && src, "null argument");
    M3D_REQUIRE(clevel >= 1 && clevel <= 9, "clevel must be 1..9");
    M3D_REQUIRE(shuffle >= 0 && shuffle <= 2, "shuffle must be 0, 1 or 2");
    M3D_REQUIRE(codec == kZSTD || codec == kLZ4, "codec must be 1 (lz4) or 4 (zstd)");
    Grid g;
    if (int rc = make_grid(shape, chunk, T, &g)) return rc;
    const auto t0 = clk::now();
    const std::string dir(chunk_dir);
    size_t blocksize = std::min<size_t>(kDefaultBlock, (size_t)g.chunk_bytes);
    for (int64_t cz = 0; cz < g.n[0]; ++cz)
        for (int64_t cy = 0; cy < g.n[1]; ++cy) make_dirs(dir + "/" + std::to_string(cz) + "/" + std::to_string(cy));

    if (device) {
        if (int rc = ensure_staging(io, (size_t)(g.nchunks * g.chunk_bytes), true)) return rc;
        for (int64_t c = 0; c < g.nchunks; ++c) {
            ChunkJob j = job_for(g, c);
            j.blocksize = (uint32_t)blocksize;
            j.shuffle = shuffle;
            uint8_t *d = io->dev + (size_t)c * g.chunk_bytes;
            switch (T) {
            case 1: launch_shuffle<1>((const uint8_t *)src, d, j, st); break;
            case 2: launch_shuffle<2>((const uint8_t *)src, d, j, st); break;
            case 4: launch_shuffle<4>((const uint8_t *)src, d, j, st); break;
            default: launch_shuffle<8>((const uint8_t *)src, d, j, st); break;
            }
        }
        M3D_KERNEL_CHECK();
        M3D_HIP_CHECK(hipMemcpyAsync(io->pinned, io->dev, (size_t)(g.nchunks * g.chunk_bytes), hipMemcpyDeviceToHost, st));
        M3D_HIP_CHECK(hipStreamSynchronize(st));
    }

    std::vector<double> busy((size_t)io->nthreads, 0.0);
    std::vector<int> codes((size_t)io->nthreads, 0);
    std::vector<std::string> errors((size_t)io->nthreads);
    std::atomic<int64_t> file_bytes{0};
    const size_t bound = m3d_blosc_bound((size_t)g.chunk_bytes, blocksize);
    m3d_blosc_header hs{};
    hs.typesize = T, hs.shuffle = shuffle;
    run_workers(io->nthreads, g.nchunks, [&](int64_t c, int t) {
        const auto w0 = clk::now();
        thread_local std::vector<uint8_t> raw, shuffled, frame, scratch;
        if (frame.size() < bound) frame.resize(bound);
        const uint8_t *body;
        if (device) {
            body = io->pinned + (size_t)c * g.chunk_bytes;
        } else {
            if (raw.size() < (size_t)g.chunk_bytes) raw.resize((size_t)g.chunk_bytes), shuffled.resize((size_t)g.chunk_bytes);
            host_gather(g, c, (const uint8_t *)src, raw.data());
            for (size_t off = 0; off < (size_t)g.chunk_bytes; off += blocksize) {
                const uint32_t bsize = (uint32_t)std::min(blocksize, (size_t)g.chunk_bytes - off);
                host_shuffle(raw.data() + off, shuffled.data() + off, bsize, T, block_shuffle(hs, bsize));
            }
            body = shuffled.data();
        }
        size_t n = 0;
        int rc = encode_frame(body, (size_t)g.chunk_bytes, T, clevel, shuffle, codec, blocksize, frame.data(), bound, &n,
                              scratch);
        int64_t cz, cy, cx;
        chunk_coords(g, c, &cz, &cy, &cx);
        if (!rc && !write_file(chunk_path(dir, cz, cy, cx), frame.data(), n)) {
            m3d::set_err("write failed", chunk_path(dir, cz, cy, cx).c_str());
            rc = M3D_ERR_INVALID_ARG;
        }
        if (rc) {
            codes[(size_t)t] = rc;
            errors[(size_t)t] = m3d::err_buf();
        }
        file_bytes += (int64_t)n;
        busy[(size_t)t] += since(w0);
    });
    int rc = M3D_OK;
    for (int t = 0; t < io->nthreads; ++t)
        if (codes[(size_t)t]) {
            rc = codes[(size_t)t];
            snprintf(m3d::err_buf(), 512, "%s", errors[(size_t)t].c_str());
            break;
        }
    if (stats) {
        stats->seconds_total = since(t0);
        stats->seconds_entropy = 0;
        for (double b : busy) stats->seconds_entropy += b;
        stats->file_bytes = file_bytes.load();
        stats->raw_bytes = g.shape[0] * g.shape[1] * g.shape[2] * T;
        stats->chunks = (int32_t)g.nchunks;
        stats->chunks_missing = 0;
    }
    return rc;
}

int m3d_zarr_write_device(m3d_io *io, const char *chunk_dir, const int64_t shape[3], const int64_t chunk[3], int typesize,
                          const void *src, int clevel, int shuffle, int codec, void *stream, m3d_io_stats *stats)
{
    return write_impl(io, chunk_dir, shape, chunk, typesize, src, clevel, shuffle, codec, true, (hipStream_t)stream, stats);
}

int m3d_zarr_write_host(m3d_io *io, const char *chunk_dir, const int64_t shape[3], const int64_t chunk[3], int typesize,
                        const void *src, int clevel, int shuffle, int codec, m3d_io_stats *stats)
{
    return write_impl(io, chunk_dir, shape, chunk, typesize, src, clevel, shuffle, codec, false, nullptr, stats);
}

}  // extern "C"
