// hfile_min.cpp -- a small local-file / file-descriptor provider of htslib's exported hFILE
// entry points (include/hts_hfile_abi.h), so that libhts_bgzf.so is usable on its own.
//
// hFILE transports (hfile.c, hfile_*.c, plugins) are NOT part of the accelerated path; inside a
// real libhts build this file is left out and the BGZF front-end binds to hfile.c instead
// (oracle/Makefile builds that configuration for the tests).  Written from the interface
// contract (htslib/hfile.h, hfile_internal.h:65-113): one buffer per stream, [begin,end) = unread
// bytes of a reader, [buffer,begin) = unwritten bytes of a writer, `offset` = stream position of
// `buffer`.  Supported names: plain paths, "-" (stdin / stdout), "file:" URLs.
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include "hts_hfile_abi.h"

namespace {

struct FdFile {
    hFILE base;
    int fd;
    bool shared;              // do not close(2) the descriptor (stdin/stdout, mode 'S')
};

bool is_writer(const hFILE *fp) { return !fp->readonly; }

// Push out [buffer, begin) of a writer.
int drain_writes(hFILE *fp) {
    const char *p = fp->buffer;
    while (p < fp->begin) {
        ssize_t n = fp->backend->write(fp, p, (size_t)(fp->begin - p));
        if (n < 0) { fp->has_errno = errno; return -1; }
        p += n; fp->offset += n;
    }
    fp->begin = fp->buffer;
    return 0;
}

// Top the read buffer up by one backend read; consumed bytes are dropped first.
ssize_t top_up(hFILE *fp) {
    if (fp->mobile && fp->begin > fp->buffer) {
        const size_t keep = (size_t)(fp->end - fp->begin);
        fp->offset += fp->begin - fp->buffer;
        memmove(fp->buffer, fp->begin, keep);
        fp->begin = fp->buffer; fp->end = fp->buffer + keep;
    }
    if (fp->at_eof || fp->end == fp->limit) return 0;
    ssize_t n = fp->backend->read(fp, fp->end, (size_t)(fp->limit - fp->end));
    if (n < 0) { fp->has_errno = errno; return n; }
    if (n == 0) fp->at_eof = 1;
    fp->end += n;
    return n;
}

ssize_t fd_read(hFILE *f, void *buf, size_t n) {
    FdFile *fp = (FdFile *)f;
    ssize_t r;
    do r = read(fp->fd, buf, n); while (r < 0 && errno == EINTR);
    return r;
}
ssize_t fd_write(hFILE *f, const void *buf, size_t n) {
    FdFile *fp = (FdFile *)f;
    ssize_t r;
    do r = write(fp->fd, buf, n); while (r < 0 && errno == EINTR);
    return r;
}
off_t fd_seek(hFILE *f, off_t off, int whence) { return lseek(((FdFile *)f)->fd, off, whence); }
int fd_flush(hFILE *) { return 0; }
int fd_close(hFILE *f) {
    FdFile *fp = (FdFile *)f;
    if (fp->shared) return 0;
    int r;
    do r = close(fp->fd); while (r < 0 && errno == EINTR);
    return r;
}
const hFILE_backend kFdBackend = {fd_read, fd_write, fd_seek, fd_flush, fd_close};

size_t buffer_hint(int fd) {
    struct stat st;
    size_t sz = 32768;
    if (fstat(fd, &st) == 0 && st.st_blksize > 0 && (size_t)st.st_blksize > sz) sz = (size_t)st.st_blksize;
    return sz > (1u << 20) ? (1u << 20) : sz;
}

hFILE *wrap_fd(int fd, const char *mode, bool shared) {
    FdFile *fp = (FdFile *)hfile_init(sizeof(FdFile), mode, buffer_hint(fd));
    if (!fp) return nullptr;
    fp->fd = fd; fp->shared = shared || strchr(mode, 'S') != nullptr;
    fp->base.backend = &kFdBackend;
    return &fp->base;
}

}  // namespace

extern "C" {

int hfile_oflags(const char *mode) {
    int access = 0, extra = 0;
    for (const char *m = mode; *m && *m != ':'; m++) {
        if (*m == 'r') access = O_RDONLY;
        else if (*m == 'w') { access = O_WRONLY; extra |= O_CREAT | O_TRUNC; }
        else if (*m == 'a') { access = O_WRONLY; extra |= O_CREAT | O_APPEND; }
        else if (*m == '+') access = O_RDWR;
        else if (*m == 'e') extra |= O_CLOEXEC;
        else if (*m == 'x') extra |= O_EXCL;
    }
    return access | extra;
}

hFILE *hfile_init(size_t struct_size, const char *mode, size_t capacity) {
    hFILE *fp = (hFILE *)calloc(1, struct_size);
    if (!fp) return nullptr;
    if (capacity == 0) capacity = 32768;
    fp->buffer = (char *)malloc(capacity);
    if (!fp->buffer) { free(fp); return nullptr; }
    fp->begin = fp->end = fp->buffer;
    fp->limit = fp->buffer + capacity;
    fp->mobile = 1;
    fp->readonly = strchr(mode, 'r') && !strchr(mode, '+');
    return fp;
}

void hfile_destroy(hFILE *fp) {
    const int keep = errno;
    if (fp) free(fp->buffer);
    free(fp);
    errno = keep;
}

int hfile_set_blksize(hFILE *fp, size_t bufsiz) {
    if (!fp) return -1;
    const size_t used = (size_t)(fp->end - fp->buffer), at = (size_t)(fp->begin - fp->buffer);
    if (bufsiz < used) bufsiz = used;
    if (bufsiz == 0) bufsiz = 32768;
    char *nb = (char *)realloc(fp->buffer, bufsiz);
    if (!nb) return -1;
    fp->buffer = nb; fp->begin = nb + at; fp->end = nb + used; fp->limit = nb + bufsiz;
    return 0;
}

hFILE *hdopen(int fd, const char *mode) { return wrap_fd(fd, mode, false); }

hFILE *hopen(const char *name, const char *mode, ...) {
    if (!name || !mode) { errno = EINVAL; return nullptr; }
    if (strcmp(name, "-") == 0) return wrap_fd(strchr(mode, 'r') ? STDIN_FILENO : STDOUT_FILENO, mode, true);
    if (strncmp(name, "file://localhost/", 17) == 0) name += 16;
    else if (strncmp(name, "file:///", 8) == 0) name += 7;
    else if (strncmp(name, "file:", 5) == 0 && name[5] != '/') name += 5;
    else if (strstr(name, "://")) { errno = EPROTONOSUPPORT; return nullptr; }
    const int fd = open(name, hfile_oflags(mode), 0666);
    if (fd < 0) return nullptr;
    hFILE *fp = wrap_fd(fd, mode, false);
    if (!fp) { const int e = errno; close(fd); errno = e; }
    return fp;
}

ssize_t hpeek(hFILE *fp, void *out, size_t n) {
    while ((size_t)(fp->end - fp->begin) < n) {
        ssize_t r = top_up(fp);
        if (r < 0) return r;
        if (r == 0) break;                     // EOF, or the buffer is full
    }
    size_t have = (size_t)(fp->end - fp->begin);
    if (have > n) have = n;
    memcpy(out, fp->begin, have);
    return (ssize_t)have;
}

int hgetc2(hFILE *fp) {
    if (fp->begin == fp->end && top_up(fp) <= 0) return -1;
    return (unsigned char)*fp->begin++;
}

// The caller has already taken every buffered byte (`done` of `want` are in dst).
ssize_t hread2(hFILE *fp, void *dstv, size_t want, size_t done) {
    char *dst = (char *)dstv;
    const size_t cap = (size_t)(fp->limit - fp->buffer);
    bool bypassed = false;
    while (want - done >= cap / 2 && !fp->at_eof) {          // big requests skip the buffer
        ssize_t n = fp->backend->read(fp, dst + done, want - done);
        if (n < 0) { fp->has_errno = errno; return n; }
        if (n == 0) fp->at_eof = 1;
        else { bypassed = true; fp->offset += n; done += (size_t)n; }
    }
    if (bypassed) {                                           // the consumed buffer no longer abuts the position
        fp->offset += fp->begin - fp->buffer;
        fp->begin = fp->end = fp->buffer;
    }
    while (done < want) {
        if (fp->begin == fp->end) {
            ssize_t r = top_up(fp);
            if (r < 0) return r;
            if (r == 0) break;
        }
        size_t take = (size_t)(fp->end - fp->begin);
        if (take > want - done) take = want - done;
        memcpy(dst + done, fp->begin, take);
        fp->begin += take; done += take;
    }
    return (ssize_t)done;
}

int hflush(hFILE *fp) {
    if (is_writer(fp) && drain_writes(fp) < 0) return -1;
    if (fp->backend->flush && fp->backend->flush(fp) < 0) { fp->has_errno = errno; return -1; }
    return 0;
}

ssize_t hwrite2(hFILE *fp, const void *srcv, size_t total, size_t copied) {
    const char *src = (const char *)srcv;
    if (drain_writes(fp) < 0) return -1;
    const size_t cap = (size_t)(fp->limit - fp->buffer);
    while (total - copied >= cap / 2 && total > copied) {     // big writes go straight out
        ssize_t n = fp->backend->write(fp, src + copied, total - copied);
        if (n < 0) { fp->has_errno = errno; return n; }
        fp->offset += n; copied += (size_t)n;
    }
    memcpy(fp->begin, src + copied, total - copied);
    fp->begin += total - copied;
    return (ssize_t)total;
}

off_t hseek(hFILE *fp, off_t off, int whence) {
    off_t here = hg_htell(fp);
    if (is_writer(fp)) {
        if (drain_writes(fp) < 0) return -1;
    } else if (whence == SEEK_CUR) { off += here; whence = SEEK_SET; }
    if (!is_writer(fp) && whence == SEEK_SET && off >= fp->offset && off <= fp->offset + (fp->end - fp->buffer)) {
        fp->begin = fp->buffer + (off - fp->offset);          // still inside the buffer
        return off;
    }
    off_t at = fp->backend->seek(fp, off, whence);
    if (at < 0) { fp->has_errno = errno; return at; }
    fp->offset = at;
    fp->begin = fp->end = fp->buffer;
    fp->at_eof = 0;
    return at;
}

int hclose(hFILE *fp) {
    int err = fp->has_errno;
    if (is_writer(fp) && hflush(fp) < 0) err = fp->has_errno;
    if (fp->backend->close(fp) < 0) err = errno;
    hfile_destroy(fp);
    if (err) { errno = err; return -1; }
    return 0;
}

void hclose_abruptly(hFILE *fp) {
    const int keep = errno;
    (void)fp->backend->close(fp);
    hfile_destroy(fp);
    errno = keep;
}

}  // extern "C"
