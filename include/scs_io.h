/*
 * scs_io.h - C ABI of libscs_io.so: host-side file helpers of the inference path of the MI355X implementation of
 * sergivalverde/sub-cortical_segmentation.  No device code and no HIP dependency: a host that only reads volumes (or a
 * process that has not decided on a GPU runtime yet) loads this library alone.
 *
 * Every function returns 0 on success, a negative value on failure; scs_io_last_error() gives the text (per thread).
 */
#ifndef SCS_IO_H
#define SCS_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* scs_io_last_error(void);

/* The elements of a gzip-compressed file at the given byte offsets of its INFLATED content
 * (ascending; n elements of elem_bytes <= 16 bytes each, copied to out in that order), without holding the inflated file - it
 * passes through a cache-sized window.  Concatenated gzip members are read as one stream.  Every member's ISIZE is checked,
 * its CRC-32 is not.  *inflated_bytes (when non-NULL) receives the total inflated length, and the whole file is then read;
 * with NULL the call returns as soon as the last element has been produced.
 * Replaces: nibabel's get_data() of the 15-channel prior (755 MB inflated per 256x256x192 subject) in front of the per-voxel
 * prior rows of load_patch_batch (base.py:388-394): 0.6 -> 0.2 CPU-seconds per subject against the system zlib. */
int scs_gz_gather(const char* path, const int64_t* byte_offsets, int64_t n, int elem_bytes, void* out, int64_t* inflated_bytes);

/* Same inflate, contiguous: bytes [offset, offset + nbytes) of the inflated content of a gzip file -> out (the voxel block of a
 * T1 or mask volume behind its 352-byte header: nifti.load). */
int scs_gz_read(const char* path, int64_t offset, int64_t nbytes, void* out);

/* One complete gzip member (header, deflate stream, CRC-32, length) holding data[0 .. nbytes), written for SPARSE volumes: zero
 * runs cost 13 bits per 258 bytes, bytes that repeat the bytes 8 back (float64 values that came from float32, constant fills)
 * are matches, everything else literals, one fixed-Huffman block.  Any inflater reads it (nibabel, zlib, gunzip); on dense
 * data it does not compress (use zlib there).  cap >= nbytes + nbytes / 8 + 64 always suffices; *out_len receives the size.
 * Replaces: the gzip stage of nibabel's save() for the probability / label volumes of test_scan (base.py:441-458): 330 MB of
 * rows of which 39 MB are data per 256x256x192 subject - zlib level 1 spends 0.38 CPU-seconds on them, this 0.1. */
int scs_gz_member(const void* data, int64_t nbytes, void* out, int64_t cap, int64_t* out_len);

/* The same member for a buffer described sparsely: nbytes of `fill` elements (item <= 16 bytes each) with values[k] written at
 * element elem_index[k] (later duplicates win).  The buffer is built inside the call - the writer threads of nifti.save_sparse
 * spend no interpreter time on it.  A cap too small for the result is an error (scs_io_last_error says so): retry with
 * nbytes + nbytes / 8 + 64. */
int scs_gz_member_scatter(int64_t nbytes, const void* fill, int item, const int64_t* elem_index, const void* values, int64_t nvals,
                          void* out, int64_t cap, int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* SCS_IO_H */
