/* m3d_io.h - C ABI of the datastore image I/O fast path (exported by libm3d_hip.so).
 *
 * Replaces, for the images the hot path reads and writes, the reference's
 * tensorstore / yaozarrs round trip (qi2labDataStore.py:2234-2373
 * `_load_from_zarr_array` / `_save_to_zarr_array`, array creation `:1425-1527`):
 * OME-Zarr 0.5 groups whose level-0 array is a Zarr v3 regular chunk grid
 * (default 16x512x512, `_default_chunks` :1562-1612), codecs `bytes` (little endian) +
 * `blosc` (cname zstd | lz4, clevel 5, shuffle bitshuffle), chunk keys `c/<z>/<y>/<x>`.
 * The on-disk bytes stay readable by the reference (any Blosc 1 decoder) and the
 * reference's files are readable here.
 *
 * Division of labour on MI355X: the entropy stage (zstd / lz4, byte-serial) runs on host
 * threads, block by block, straight into pinned staging; the bit/byte (un)shuffle and the
 * chunk <-> volume scatter/gather run as HIP kernels on data that crosses PCIe still
 * shuffled, so the host never touches a decoded voxel.  The *_host entry points do the
 * whole job on the CPU (validation, hosts without a GPU, and the baseline timed beside it).
 *
 * Conventions: as m3d_hip.h (status codes, m3d_last_error).  Shapes are int64[3] Z,Y,X;
 * 4-D / 5-D arrays with leading chunk extent 1 are read one leading index at a time by
 * passing `<array>/c/<i>` as chunk_dir.  zstd / lz4 come from the system's libzstd.so.1 /
 * liblz4.so.1, loaded on first use (M3D_ERR_UNSUPPORTED when absent).
 */
#ifndef M3D_IO_H
#define M3D_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Blosc 1 frame header (16 bytes on disk), decoded. */
typedef struct m3d_blosc_header {
    uint32_t nbytes;     /* uncompressed size                                  */
    uint32_t blocksize;  /* bytes per block (last block may be shorter)        */
    uint32_t cbytes;     /* frame size including the header                    */
    int32_t typesize;
    int32_t shuffle;     /* 0 none, 1 byte shuffle, 2 bit shuffle              */
    int32_t codec;       /* 0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd */
    int32_t memcpyed;    /* payload stored verbatim after the header           */
    int32_t split;       /* blocks are stored as `typesize` streams            */
    int32_t nblocks;
} m3d_blosc_header;

int m3d_blosc_info(const void *frame_host, size_t frame_bytes, m3d_blosc_header *out);
/* Full decode on the host (entropy stage + unshuffle).  dst_bytes must equal header.nbytes. */
int m3d_blosc_decode_host(const void *frame_host, size_t frame_bytes, void *dst_host, size_t dst_bytes);
/* Upper bound of an encoded frame. */
size_t m3d_blosc_bound(size_t nbytes, size_t blocksize);
/* Encode on the host: shuffle 0/1/2, codec 1 (lz4) or 4 (zstd), clevel 1..9 (the Blosc
 * mapping zstd level = 2*clevel-1), blocksize 0 = 256 KiB.  Blocks are never split. */
int m3d_blosc_encode_host(const void *src_host, size_t nbytes, int typesize, int clevel, int shuffle, int codec,
                          size_t blocksize, void *dst_host, size_t dst_capacity, size_t *out_bytes);

/* Reader / writer state: host worker threads, pinned staging, device staging (grown on demand). */
typedef struct m3d_io m3d_io;
int m3d_io_create(int nthreads, m3d_io **out);
void m3d_io_destroy(m3d_io *io);

typedef struct m3d_io_stats {
    double seconds_total;    /* wall time of the call                                   */
    double seconds_entropy;  /* summed over worker threads: file I/O + zstd / lz4       */
    int64_t file_bytes;      /* compressed bytes read or written                        */
    int64_t raw_bytes;       /* voxel bytes delivered / consumed                        */
    int32_t chunks;
    int32_t chunks_missing;  /* read: absent chunk files (filled with zero, Zarr rule)  */
} m3d_io_stats;

/* Read every chunk `chunk_dir/<z>/<y>/<x>` of a shape_zyx array with chunk_zyx chunks of
 * `typesize`-byte little-endian voxels into a C-contiguous volume.
 *  _device: dst is a DEVICE pointer; entropy stage on host threads, unshuffle + scatter on
 *           `stream`; the call returns after the stream has drained (staging is reused).
 *  _host:   dst is a host pointer; everything on host threads.                          */
int m3d_zarr_read_device(m3d_io *io, const char *chunk_dir, const int64_t shape_zyx[3], const int64_t chunk_zyx[3],
                         int typesize, void *dst, void *stream, m3d_io_stats *stats);
int m3d_zarr_read_host(m3d_io *io, const char *chunk_dir, const int64_t shape_zyx[3], const int64_t chunk_zyx[3],
                       int typesize, void *dst_host, m3d_io_stats *stats);

/* Write a volume as chunk files (edge chunks zero-padded to the full chunk shape, as
 * tensorstore writes them).  Directories below chunk_dir are created.
 *  _device: src is a DEVICE pointer; gather + shuffle on `stream`, entropy stage on host.  */
int m3d_zarr_write_device(m3d_io *io, const char *chunk_dir, const int64_t shape_zyx[3], const int64_t chunk_zyx[3],
                          int typesize, const void *src, int clevel, int shuffle, int codec, void *stream,
                          m3d_io_stats *stats);
int m3d_zarr_write_host(m3d_io *io, const char *chunk_dir, const int64_t shape_zyx[3], const int64_t chunk_zyx[3],
                        int typesize, const void *src_host, int clevel, int shuffle, int codec, m3d_io_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* M3D_IO_H */
