/* cram_reader_front.c -- the whole-slice CRAM path UNDER sam_read1: cram_get_bam_seq (reference cram/cram_decode.c:3615-3627) served by the device.
 *
 * The reference decodes a CRAM record by record on the host: cram_get_bam_seq -> cram_get_seq -> cram_next_slice (cram_decode.c:3268-3538: container and
 * slice I/O, one cram_decode_slice job per slice on the thread pool) -> cram_to_bam per record.  Putting the GPU under the per-block entry points
 * (cram_uncompress_block, cram_block_front.cpp) leaves that structure in place and loses 5-7x to stock htslib (profiles/r06_libhts_cram_view.txt): what the
 * device has is breadth, and the reference's reader looks two slices per thread ahead.  This file puts the GPU where the breadth is: a producer thread
 * reads a RUN of containers (hundreds of slices: cram_read_container for each header -- the reference's own function, so the EOF / error state of the
 * cram_fd is the reference's -- and one hread for each body), hands the bodies to hg_cram_containers_to_bam_host (every block of the run through the block
 * codecs in one batch, every slice through cram_decode_slice + cram_to_bam on the device, htsgpu.h) and the consumer -- the caller's thread inside
 * cram_get_bam_seq -- copies one BAM record per call into the caller's bam1_t, the way bam_read1 lays a record out in memory (sam.c:784-866).
 *
 * This is an INTEGRATION source: it is compiled against htslib's private headers (cram/cram.h) inside a libhts build, as a maintainer would add it
 * (INTEGRATION.md A3); oracle/Makefile builds it into oracle/_ref/libhts_gpu.so, where the reference's cram_get_bam_seq / cram_seek / cram_close are
 * renamed hg_ref_* (objcopy --redefine-sym) and the functions below take their names.  libhtsgpu.so and libhts_bgzf.so do not contain it.
 *
 * When the reference's own path runs instead (always complete, never partial):
 *   - a region is set (fd->range.refid != -2: iterators, CRAM_OPT_RANGE), required_fields was narrowed, the file is CRAM 1.x / 4.x, the input cannot
 *     seek, HTS_GPU_CRAM_SLICE=0, or no device context -- decided when the first record is asked for;
 *   - ANY failure inside a run (truncated body, CRC or MD5 mismatch, a slice the device decoder declines, a missing reference): the file is seeked back to
 *     the first container of that run and the reference's decoder takes over from there for good, so records before the fault come out as they would, the
 *     fault is reported by the reference's own code with its own message and return value.
 * Runs are handed over whole: no record of a run is delivered before the run has decoded, so the switch is always at a container boundary. */
#include <config.h>

#include <errno.h>
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cram/cram.h"
#include "htslib/hfile.h"
#include "htslib/hts_log.h"
#include "htslib/sam.h"

#include "htsgpu.h"

/* the reference's bodies under their new names (oracle/Makefile: REDEF_*) */
int hg_ref_cram_get_bam_seq(cram_fd *fd, bam_seq_t **bam);
int hg_ref_cram_seek(cram_fd *fd, off_t offset, int whence);
int hg_ref_cram_close(cram_fd *fd);
int sam_realloc_bam_data(bam1_t *b, size_t desired);         /* sam_internal.h:45 */
/* the block layer's device context (cram_block_front.cpp) */
hg_ctx *hg_front_shared_engine(void);

enum { RUN_RECORDS, RUN_END, RUN_FALLBACK };

typedef struct run {
    int kind;                      /* RUN_RECORDS: bam[0 .. bam_len) holds nrec records; RUN_END: the input ended (fd_eof / fd_err = the cram_fd's state at
                                      that point); RUN_FALLBACK: the reference's decoder continues at start_off */
    off_t start_off;               /* file offset of the run's first container header */
    uint8_t *raw; size_t raw_cap;  /* the container bodies, back to back */
    hg_cram_container *cont; size_t ncont, cont_cap;
    uint8_t *bam; size_t bam_cap; uint64_t bam_len, nrec;
    int fd_eof, fd_err;
} run;

typedef struct reader {
    struct reader *next;
    cram_fd *fd;
    hg_ctx *ctx;
    int pass;                      /* 1: the reference's path owns this cram_fd until the next seek */
    /* producer <-> consumer: two runs, `ready` = decoded runs waiting (0..2), `head` = the consumer's */
    pthread_t th; int th_on;
    pthread_mutex_t m; pthread_cond_t cv;
    run r[2]; int head, ready, stop, done;   /* done: the producer has delivered a RUN_END / RUN_FALLBACK and exited */
    run *cur; uint64_t pos;        /* the run being handed out and the offset of its next record */
    unsigned runs_made;
    /* header facts the decoder wants, gathered once */
    int nref, nrg; int64_t *sq_len; char **rg_names;
    int *held; int nheld, held_cap;   /* reference ids pinned by cram_get_ref for the run being decoded */
    int end_seen, end_eof, end_err;   /* cram_read_container has already said "no more": the next run is that news, it is not asked twice */
    int stats; double t_io, t_dec, t_start, t_first, t_end, t_wait; uint64_t tot_rec, tot_bam;
} reader;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static reader *g_readers;
static volatile unsigned g_gen = 1;                         /* bumped whenever a reader goes away: invalidates the per-thread cache below */
static __thread cram_fd *tl_fd; static __thread reader *tl_rd; static __thread unsigned tl_gen;

static double g_t0;
static void stats_at_exit(void);
__attribute__((constructor)) static void reader_loaded(void) {
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); g_t0 = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
    const char *e = getenv("HTS_GPU_STATS");
    if (e && e[0] == '1') atexit(stats_at_exit);
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void stats_at_exit(void) { fprintf(stderr, "[htsgpu stats] process: exit handlers reached %.3f s after libhts was loaded\n", now_s() - g_t0); }

static reader *find_reader(cram_fd *fd) {
    if (tl_fd == fd && tl_gen == g_gen) return tl_rd;
    pthread_mutex_lock(&g_lock);
    reader *r = g_readers;
    while (r && r->fd != fd) r = r->next;
    tl_fd = fd; tl_rd = r; tl_gen = g_gen;
    pthread_mutex_unlock(&g_lock);
    return r;
}

/* ---- references on demand: cram_get_ref pins a whole sequence (cram/cram_io.c:3409-3590), cram_ref_decr lets it go after the run ---- */
static int get_ref_cb(void *ud, int id, hg_cram_ref_seq *out) {
    reader *R = (reader *)ud;
    cram_fd *fd = R->fd;
    if (!fd->refs || id < 0 || id >= fd->refs->nref) return -1;
    char *seq = cram_get_ref(fd, id, 1, 0);
    if (!seq) return -1;
    if (R->nheld == R->held_cap) {
        int cap = R->held_cap ? 2 * R->held_cap : 16;
        int *h = realloc(R->held, (size_t)cap * sizeof *h);
        if (!h) { cram_ref_decr(fd->refs, id); return -1; }
        R->held = h; R->held_cap = cap;
    }
    R->held[R->nheld++] = id;
    pthread_mutex_lock(&fd->refs->lock);
    const int64_t len = fd->refs->ref_id[id]->length;
    pthread_mutex_unlock(&fd->refs->lock);
    out->bases = (const uint8_t *)seq; out->len = (uint64_t)len;
    return 0;
}
static void release_refs(reader *R) {
    for (int i = 0; i < R->nheld; i++) cram_ref_decr(R->fd->refs, R->held[i]);
    R->nheld = 0;
}

/* The two big buffers of a run are plain malloc memory, kept and reused by the run after next.  Measured on the MI355X box (experiments/pinned_probe.hip,
 * profiles/r06_pinned_probe.txt): a transfer into pageable memory that has been touched runs at the same 56 GB/s as one into page-locked memory, while
 * page-locking costs 215 ms per GiB (and 128 ms to undo) -- more than the first-touch page faults it would save (100 ms per GiB, once per buffer). */
static int big_grow(uint8_t **p, size_t *cap, size_t keep, size_t need) {
    if (need <= *cap) return 0;
    size_t c = *cap ? *cap : (size_t)64 << 20;
    while (c < need) c *= 2;
    uint8_t *q = keep ? realloc(*p, c) : malloc(c);             /* (a large realloc moves pages, not bytes) */
    if (!q) return -1;
    if (!keep) free(*p);
    *p = q; *cap = c;
    return 0;
}

/* ---- the producer: one run = container headers through the reference's cram_read_container, bodies raw, then the device ---- */
static void fill_run(reader *R, run *u, size_t want_slices) {
    cram_fd *fd = R->fd;
    const double t0 = now_s();
    u->kind = RUN_RECORDS; u->ncont = 0; u->bam_len = 0; u->nrec = 0; u->fd_eof = 0; u->fd_err = 0;
    u->start_off = htell(fd->fp);
    if (R->end_seen) { u->kind = RUN_END; u->fd_eof = R->end_eof; u->fd_err = R->end_err; return; }
    size_t raw_len = 0, slices = 0;
    uint64_t bases = 0, records = 0;
    int ended = 0;
    /* bodies are addressed by offset until the buffer stops moving */
    size_t *body_off = NULL; size_t off_cap = 0;
    while (slices < want_slices && raw_len < ((size_t)3 << 29)) {
        cram_container *c = cram_read_container(fd);
        if (!c) {                                              /* end of input, or a malformed header: either way the reference's verdict is in fd->eof / fd->err */
            ended = 1; u->fd_eof = R->end_eof = fd->eof; u->fd_err = R->end_err = fd->err; R->end_seen = 1;
            break;
        }
        const int32_t len = c->length, nrec = c->num_records, nblk = c->num_blocks, nland = c->num_landmarks;
        const int64_t nb = c->num_bases;
        cram_free_container(c);
        if (len < 0) { u->kind = RUN_FALLBACK; break; }
        if (len == 0) continue;
        if (big_grow(&u->raw, &u->raw_cap, raw_len, raw_len + (size_t)len) < 0) { u->kind = RUN_FALLBACK; break; }
        if (hread(fd->fp, u->raw + raw_len, (size_t)len) != (ssize_t)len) { u->kind = RUN_FALLBACK; break; }   /* truncated: the reference says how */
        if (nrec == 0) continue;                                /* the EOF container and other empty ones (cram_decode.c:3404-3414) */
        if (u->ncont == u->cont_cap) {
            size_t cap = u->cont_cap ? 2 * u->cont_cap : 256;
            hg_cram_container *p = realloc(u->cont, cap * sizeof *p);
            if (!p) { u->kind = RUN_FALLBACK; break; }
            u->cont = p; u->cont_cap = cap;
        }
        if (u->ncont == off_cap) {
            size_t cap = off_cap ? 2 * off_cap : 256;
            size_t *p = realloc(body_off, cap * sizeof *p);
            if (!p) { u->kind = RUN_FALLBACK; break; }
            body_off = p; off_cap = cap;
        }
        body_off[u->ncont] = raw_len;
        u->cont[u->ncont].body = NULL; u->cont[u->ncont].body_len = (uint32_t)len; u->cont[u->ncont].num_blocks = nblk; u->cont[u->ncont].bases = nb > 0 ? (uint64_t)nb : 0;
        u->ncont++;
        raw_len += (size_t)len;
        slices += nland > 0 ? (size_t)nland : 1;
        bases += nb > 0 ? (uint64_t)nb : 0; records += (uint64_t)nrec;
    }
    for (size_t i = 0; i < u->ncont; i++) u->cont[i].body = u->raw + body_off[i];
    free(body_off);
    const double t1 = now_s();
    R->t_io += t1 - t0;
    if (u->kind == RUN_FALLBACK) return;
    if (u->ncont == 0) { u->kind = ended ? RUN_END : RUN_FALLBACK; return; }
    /* records of this run first; the end of the input is the NEXT run's news (end_seen) */
    size_t need = (size_t)(bases + bases / 2 + records * 320 + ((size_t)1 << 20));
    for (int attempt = 0; attempt < 2; attempt++) {
        if (big_grow(&u->bam, &u->bam_cap, 0, need) < 0) { u->kind = RUN_FALLBACK; return; }
        uint64_t bytes = 0, nrec = 0;
        const int rc = hg_cram_containers_to_bam_host(R->ctx, CRAM_MAJOR_VERS(fd->version), u->ncont, u->cont, R->nref, R->sq_len, (const char *const *)R->rg_names, R->nrg,
                                                      NULL, 0, get_ref_cb, R, fd->ignore_md5 ? HG_CRAM_IGNORE_MD5 : 0, fd->decode_md, fd->prefix, u->bam, u->bam_cap,
                                                      &bytes, &nrec);
        release_refs(R);
        if (rc == HG_OK) { u->bam_len = bytes; u->nrec = nrec; break; }
        if (rc == HG_ENOMEM && attempt == 0 && bytes > u->bam_cap) { need = (size_t)bytes + ((size_t)1 << 20); continue; }
        u->kind = RUN_FALLBACK;                                 /* whatever it was, the reference's decoder finds it again and names it */
        return;
    }
    if (u->nrec != records) { u->kind = RUN_FALLBACK; return; } /* a slice came back short: not ours to paper over */
    R->t_dec += now_s() - t1; R->tot_rec += u->nrec; R->tot_bam += u->bam_len;
    (void)ended;
}

static void *producer(void *arg) {
    reader *R = (reader *)arg;
    for (;;) {
        pthread_mutex_lock(&R->m);
        while (R->ready == 2 && !R->stop) pthread_cond_wait(&R->cv, &R->m);
        if (R->stop) { pthread_mutex_unlock(&R->m); break; }
        run *u = &R->r[(R->head + R->ready) & 1];
        pthread_mutex_unlock(&R->m);
        /* A run takes 0.2-0.3 s whatever it holds (its longest entropy-coded stream is one chain on one wavefront), so runs are made wide: 256 slices
         * before the first record is out, 1024 from then on -- one slice stream for every SIMD of the device. */
        const size_t want = R->runs_made == 0 ? 256 : 1024;
        fill_run(R, u, want);
        R->runs_made++;
        pthread_mutex_lock(&R->m);
        R->ready++;
        const int last = u->kind != RUN_RECORDS;
        if (last) R->done = 1;
        pthread_cond_broadcast(&R->cv);
        pthread_mutex_unlock(&R->m);
        if (last) break;
    }
    return NULL;
}

static void stop_producer(reader *R) {
    if (!R->th_on) return;
    pthread_mutex_lock(&R->m);
    R->stop = 1;
    pthread_cond_broadcast(&R->cv);
    pthread_mutex_unlock(&R->m);
    pthread_join(R->th, NULL);
    R->th_on = 0;
}

static void free_reader(reader *R) {
    stop_producer(R);
    if (R->stats)
        fprintf(stderr, "[htsgpu stats] cram reader: %u runs, %llu records, %.1f MB of BAM; producer: I/O %.3f s, device decode %.3f s; consumer: first record after %.3f s, "
                "end of input after %.3f s, waited for runs %.3f s, closed after %.3f s (reader started %.3f s after libhts was loaded)\n", R->runs_made, (unsigned long long)R->tot_rec, (double)R->tot_bam / 1e6, R->t_io, R->t_dec,
                R->t_first - R->t_start, R->t_end - R->t_start, R->t_wait, now_s() - R->t_start, R->t_start - g_t0);
    for (int i = 0; i < 2; i++) { free(R->r[i].raw); free(R->r[i].cont); free(R->r[i].bam); }
    for (int i = 0; i < R->nrg; i++) free(R->rg_names[i]);
    free(R->rg_names); free(R->sq_len); free(R->held);
    pthread_mutex_destroy(&R->m); pthread_cond_destroy(&R->cv);
    free(R);
}

/* take the cram_fd's reader off the list (seek, close): the producer is stopped first -- it owns fd->fp while it runs */
static reader *detach_reader(cram_fd *fd) {
    pthread_mutex_lock(&g_lock);
    reader **pp = &g_readers, *R = NULL;
    while (*pp && (*pp)->fd != fd) pp = &(*pp)->next;
    if (*pp) { R = *pp; *pp = R->next; g_gen++; }
    pthread_mutex_unlock(&g_lock);
    return R;
}

static int eligible(cram_fd *fd) {
    static int enabled = -1;
    if (enabled < 0) { const char *e = getenv("HTS_GPU_CRAM_SLICE"); enabled = !(e && e[0] == '0'); }
    if (!enabled || !fd || fd->mode != 'r' || !fd->fp || !fd->header) return 0;
    const int major = CRAM_MAJOR_VERS(fd->version);
    if (major != 2 && major != 3) return 0;
    if (fd->range.refid != -2 || fd->required_fields != INT_MAX) return 0;
    if (fd->ctr || fd->ctr_mt || fd->job_pending || fd->ooc) return 0;                   /* the reference's reader is in the middle of something */
    return 1;
}

static reader *start_reader(cram_fd *fd) {
    hg_ctx *ctx = hg_front_shared_engine();
    if (!ctx) return NULL;
    const off_t here = htell(fd->fp);
    if (hseek(fd->fp, here, SEEK_SET) < 0) { hclearerr(fd->fp); return NULL; }           /* a pipe: no way back for the fall-back */
    reader *R = calloc(1, sizeof *R);
    if (!R) return NULL;
    R->fd = fd; R->ctx = ctx; R->t_start = now_s();
    sam_hdr_t *h = fd->header;
    R->nref = sam_hdr_nref(h);
    R->nrg = sam_hdr_count_lines(h, "RG");
    if (R->nref < 0) R->nref = 0;
    if (R->nrg < 0) R->nrg = 0;
    R->sq_len = calloc((size_t)R->nref + 1, sizeof *R->sq_len);
    R->rg_names = calloc((size_t)R->nrg + 1, sizeof *R->rg_names);
    int ok = R->sq_len && R->rg_names;
    for (int i = 0; ok && i < R->nref; i++) R->sq_len[i] = (int64_t)sam_hdr_tid2len(h, i);
    for (int i = 0; ok && i < R->nrg; i++) {
        const char *id = sam_hdr_line_name(h, "RG", i);
        if (!id || !(R->rg_names[i] = strdup(id))) ok = 0;
    }
    pthread_mutex_init(&R->m, NULL); pthread_cond_init(&R->cv, NULL);
    { const char *e = getenv("HTS_GPU_STATS"); R->stats = e && e[0] == '1'; }
    if (!ok || pthread_create(&R->th, NULL, producer, R) != 0) { free_reader(R); return NULL; }
    R->th_on = 1;
    pthread_mutex_lock(&g_lock);
    R->next = g_readers; g_readers = R; g_gen++;
    pthread_mutex_unlock(&g_lock);
    return R;
}

/* one record of the run into the caller's bam1_t: the memory layout bam_read1 builds (sam.c:808-866; QNAME padded with NULs to a multiple of four) */
static int hand_out(reader *R, bam1_t *b) {
    run *u = R->cur;
    const uint8_t *p = u->bam + R->pos;
    if (u->bam_len - R->pos < 36) return -1;
    const uint32_t block_len = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    if (block_len < 32 || (uint64_t)block_len + 4 > u->bam_len - R->pos) return -1;
    const uint8_t *x = p + 4;
    bam1_core_t *c = &b->core;
    #define U32(q) ((uint32_t)(q)[0] | (uint32_t)(q)[1] << 8 | (uint32_t)(q)[2] << 16 | (uint32_t)(q)[3] << 24)
    c->tid = (int32_t)U32(x); c->pos = (int32_t)U32(x + 4);
    const uint32_t x2 = U32(x + 8), x3 = U32(x + 12);
    c->bin = (uint16_t)(x2 >> 16); c->qual = (uint8_t)(x2 >> 8); c->l_qname = (uint16_t)(x2 & 0xff);
    c->l_extranul = (uint8_t)((c->l_qname % 4) ? 4 - c->l_qname % 4 : 0);
    c->flag = (uint16_t)(x3 >> 16); c->n_cigar = x3 & 0xffff;
    c->l_qseq = (int32_t)U32(x + 16); c->mtid = (int32_t)U32(x + 20); c->mpos = (int32_t)U32(x + 24); c->isize = (int32_t)U32(x + 28);
    #undef U32
    const uint32_t l_name = c->l_qname, body = block_len - 32;
    if (l_name < 1 || l_name > body) return -1;
    const size_t l_data = (size_t)body + c->l_extranul;
    if (l_data > b->m_data && sam_realloc_bam_data(b, l_data) < 0) return -1;
    memcpy(b->data, x + 32, l_name);
    for (int i = 0; i < c->l_extranul; i++) b->data[l_name + (uint32_t)i] = 0;
    c->l_qname = (uint16_t)(l_name + c->l_extranul);
    memcpy(b->data + c->l_qname, x + 32 + l_name, body - l_name);
    b->l_data = (int)l_data;
    R->pos += (uint64_t)block_len + 4;
    return (int)l_data;
}

/* ---- the reference's names ---------------------------------------------------------------------------------------------------------- */
int cram_get_bam_seq(cram_fd *fd, bam_seq_t **bam) {
    reader *R = find_reader(fd);
    if (!R) {
        if (!eligible(fd) || !(R = start_reader(fd))) return hg_ref_cram_get_bam_seq(fd, bam);
        tl_fd = fd; tl_rd = R; tl_gen = g_gen;
    }
    if (R->pass) return hg_ref_cram_get_bam_seq(fd, bam);
    for (;;) {
        if (R->cur && R->pos < R->cur->bam_len) {
            if (!*bam && !(*bam = bam_init1())) return -1;
            const int n = hand_out(R, *bam);
            if (n >= 0) { if (R->t_first == 0) R->t_first = now_s(); return n; }
            hts_log_error("The device decoder returned a malformed BAM record");
            fd->err = EIO; fd->eof = 0;
            return -1;
        }
        pthread_mutex_lock(&R->m);
        if (R->cur) { R->cur = NULL; R->head ^= 1; R->ready--; pthread_cond_broadcast(&R->cv); }   /* the finished run goes back to the producer */
        const double tw = now_s();
        while (R->ready == 0) pthread_cond_wait(&R->cv, &R->m);
        run *u = &R->r[R->head];
        pthread_mutex_unlock(&R->m);
        R->t_wait += now_s() - tw;
        if (u->kind == RUN_RECORDS) { R->cur = u; R->pos = 0; continue; }
        /* the producer has exited (a run that is not records is its last) */
        stop_producer(R);
        if (u->kind == RUN_END) {
            if (R->t_end == 0) R->t_end = now_s();
            fd->eof = u->fd_eof; fd->err = u->fd_err;
            return -1;                                          /* stays RUN_END: asking again gives the same answer, as the reference's reader does */
        }
        /* RUN_FALLBACK: back to the first container of the run, the reference's reader from here on */
        if (hseek(fd->fp, u->start_off, SEEK_SET) < 0) { fd->err = errno ? errno : EIO; fd->eof = 0; return -1; }
        fd->eof = 0; fd->err = 0;
        R->pass = 1;
        return hg_ref_cram_get_bam_seq(fd, bam);
    }
}

int cram_seek(cram_fd *fd, off_t offset, int whence) {
    reader *R = detach_reader(fd);
    if (R) {
        stop_producer(R);
        if (whence == SEEK_CUR && !R->pass) {                   /* relative to where the caller believes the file is: the first container not yet handed out */
            run *u = R->cur ? R->cur : (R->ready ? &R->r[R->head] : NULL);
            if (u && hseek(fd->fp, u->start_off, SEEK_SET) < 0) { free_reader(R); return -1; }
        }
        free_reader(R);
    }
    return hg_ref_cram_seek(fd, offset, whence);
}

int cram_close(cram_fd *fd) {
    reader *R = fd ? detach_reader(fd) : NULL;
    if (R) free_reader(R);
    return hg_ref_cram_close(fd);
}
