// Feature ingest: the step before the hot path.
//
// Host side: a .npy reader that copies a ROW RANGE of a 2-D array straight into caller memory (a pinned staging buffer) as
// fp32 -- what datasets/load_features.py:50-53,67-71 does with np.load + torch.from_numpy(...).float() + slicing, without the
// intermediate arrays.  Thread-safe and GIL-free (plain C ABI, pread), so a Python thread pool reads a batch in parallel.
//
// Device side: one kernel turns the packed ragged rows of a batch (sample after sample, no padding: that is all that crosses
// PCIe) into the padded (B, T, D) tensor the model takes, filling the tail of every sample with the pad value -- the result of
// pad_sequence(..., padding_value) (datasets/captioning_dataset.py:259-261) or pad_segment (datasets/load_features.py:38-44).
// HBM-bound: reads the packed rows once, writes the padded batch once.
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include "common.h"

namespace {

struct NpyInfo {
    int64_t rows, cols, data_off;
    int elem;  // 4: <f4, 8: <f8
};

// parses the header of a version 1.x / 2.x / 3.x .npy file holding a C-ordered little-endian float array of rank 1 or 2
int npy_parse(int fd, const char* path, NpyInfo* out) {
    unsigned char pre[12];
    if (pread(fd, pre, 12, 0) < 10 || memcmp(pre, "\x93NUMPY", 6) != 0) {
        bmt_set_error("bmt_npy: %s is not a .npy file", path);
        return BMT_EINVAL;
    }
    const int major = pre[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = (size_t)pre[8] | ((size_t)pre[9] << 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        hlen = (size_t)pre[8] | ((size_t)pre[9] << 8) | ((size_t)pre[10] << 16) | ((size_t)pre[11] << 24);
        hoff = 12;
    } else {
        bmt_set_error("bmt_npy: %s: unsupported .npy version %d", path, major);
        return BMT_EINVAL;
    }
    if (hlen == 0 || hlen > 65536) {
        bmt_set_error("bmt_npy: %s: bad header length %zu", path, hlen);
        return BMT_EINVAL;
    }
    char* h = (char*)malloc(hlen + 1);
    if (!h || pread(fd, h, hlen, (off_t)hoff) != (ssize_t)hlen) {
        free(h);
        bmt_set_error("bmt_npy: %s: short header", path);
        return BMT_EINVAL;
    }
    h[hlen] = 0;
    int rc = BMT_OK;
    const char* d = strstr(h, "'descr'");
    const char* f = strstr(h, "'fortran_order'");
    const char* s = strstr(h, "'shape'");
    out->elem = 0;
    if (d) {
        if (strstr(d, "'<f4'") && strstr(d, "'<f4'") < d + 24) out->elem = 4;
        else if (strstr(d, "'<f8'") && strstr(d, "'<f8'") < d + 24) out->elem = 8;
    }
    if (!d || !f || !s || out->elem == 0) {
        bmt_set_error("bmt_npy: %s: only little-endian float32 / float64 arrays are supported (header: %.80s)", path, h);
        rc = BMT_EINVAL;
    } else if (strstr(f, "True") && strstr(f, "True") < f + 24) {
        bmt_set_error("bmt_npy: %s: fortran_order arrays are not supported", path);
        rc = BMT_EINVAL;
    } else {
        const char* p = strchr(s, '(');
        int64_t dims[3] = {0, 0, 0};
        int nd = 0;
        if (p) {
            ++p;
            while (*p && *p != ')') {
                while (*p == ' ' || *p == ',') ++p;
                if (*p == ')' || !*p) break;
                char* e;
                const long long v = strtoll(p, &e, 10);
                if (e == p || nd == 3) { nd = 99; break; }
                dims[nd++] = v;
                p = e;
            }
        }
        if (!p || nd < 1 || nd > 2) {
            bmt_set_error("bmt_npy: %s: expected a 1-D or 2-D array", path);
            rc = BMT_EINVAL;
        } else {
            out->rows = dims[0];
            out->cols = (nd == 2) ? dims[1] : 1;
            out->data_off = (int64_t)(hoff + hlen);
        }
    }
    free(h);
    return rc;
}

int read_all(int fd, void* dst, size_t n, off_t off) {
    char* p = (char*)dst;
    while (n) {
        const ssize_t r = pread(fd, p, n, off);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return -1;
        p += r; off += r; n -= (size_t)r;
    }
    return 0;
}

// packed[offsets[b] + t][:] (t < len_b) -> out[b][t][:], pad beyond; VEC floats per thread (4: one 16-byte segment)
template <int VEC>
__global__ __launch_bounds__(256) void pad_batch_kernel(const float* __restrict__ packed, const int64_t* __restrict__ offsets,
                                                        int T, int D, int64_t total, float pad, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // segment index over (b, t, c / VEC)
    if (i >= total) return;
    const int per = D / VEC;
    const int c = (int)(i % per) * VEC;
    const int64_t bt = i / per;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int64_t o0 = offsets[b];
    const bool real = t < (int)(offsets[b + 1] - o0);
    float* o = out + bt * D + c;
    if constexpr (VEC == 4) {
        *(float4*)o = real ? *(const float4*)(packed + (o0 + t) * D + c) : make_float4(pad, pad, pad, pad);
    } else {
        *o = real ? packed[(o0 + t) * D + c] : pad;
    }
}

}  // namespace

extern "C" int bmt_npy_shape(const char* path, int64_t* rows, int64_t* cols, int* elem_bytes) {
    BMT_CHECK_ARG(path && rows && cols, "bmt_npy_shape: null argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        bmt_set_error("bmt_npy_shape: cannot open %s: %s", path, strerror(errno));
        return BMT_ENOENT;
    }
    NpyInfo ni;
    const int rc = npy_parse(fd, path, &ni);
    close(fd);
    if (rc != BMT_OK) return rc;
    *rows = ni.rows;
    *cols = ni.cols;
    if (elem_bytes) *elem_bytes = ni.elem;
    return BMT_OK;
}

extern "C" int bmt_npy_read_rows(const char* path, int64_t row0, int64_t row1, float* dst, int64_t dst_floats, int64_t* rows,
                                 int64_t* cols) {
    BMT_CHECK_ARG(path && rows && cols, "bmt_npy_read_rows: null argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        bmt_set_error("bmt_npy_read_rows: cannot open %s: %s", path, strerror(errno));
        return BMT_ENOENT;
    }
    NpyInfo ni;
    int rc = npy_parse(fd, path, &ni);
    if (rc != BMT_OK) {
        close(fd);
        return rc;
    }
    if (row1 < 0 || row1 > ni.rows) row1 = ni.rows;       // row1 < 0: to the end
    if (row0 < 0) row0 = 0;
    if (row0 > row1) row0 = row1;
    const int64_t n = (row1 - row0) * ni.cols;
    *rows = row1 - row0;
    *cols = ni.cols;
    if (n == 0 || dst == nullptr) {                        // shape query of the range
        close(fd);
        return BMT_OK;
    }
    if (n > dst_floats) {
        close(fd);
        bmt_set_error("bmt_npy_read_rows: %s rows [%lld, %lld) need %lld floats, destination holds %lld", path, (long long)row0,
                      (long long)row1, (long long)n, (long long)dst_floats);
        return BMT_EINVAL;
    }
    const off_t off = (off_t)(ni.data_off + row0 * ni.cols * ni.elem);
    if (ni.elem == 4) {
        rc = read_all(fd, dst, (size_t)n * 4, off);
    } else {   // float64 on disk -> float32 (what .float() does)
        double* tmp = (double*)malloc((size_t)n * 8);
        rc = tmp ? read_all(fd, tmp, (size_t)n * 8, off) : -1;
        if (rc == 0)
            for (int64_t i = 0; i < n; ++i) dst[i] = (float)tmp[i];
        free(tmp);
    }
    close(fd);
    if (rc != 0) {
        bmt_set_error("bmt_npy_read_rows: %s: short read (file truncated?)", path);
        return BMT_EINVAL;
    }
    return BMT_OK;
}

extern "C" int bmt_pad_batch(const float* packed, const int64_t* offsets, int B, int T, int D, float pad, float* out, void* stream) {
    BMT_CHECK_ARG(packed && offsets && out, "bmt_pad_batch: null pointer");
    BMT_CHECK_ARG(B > 0 && T > 0 && D > 0, "bmt_pad_batch: B=%d T=%d D=%d out of range", B, T, D);
    const bool vec = (D & 3) == 0 && (((uintptr_t)packed | (uintptr_t)out) & 15) == 0;
    const int64_t total = (int64_t)B * T * (vec ? D / 4 : D);
    BMT_CHECK_ARG(total / 256 < 0x7fffffff, "bmt_pad_batch: batch too large for one launch");
    if (vec) pad_batch_kernel<4><<<bmt_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(packed, offsets, T, D, total, pad, out);
    else pad_batch_kernel<1><<<bmt_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(packed, offsets, T, D, total, pad, out);
    BMT_CHECK_LAUNCH("bmt_pad_batch");
    return BMT_OK;
}
