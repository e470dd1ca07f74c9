/* cram_record_front.c -- the whole-slice CRAM path UNDER sam_read1 / sam_write1: cram_get_bam_seq (reference cram/cram_decode.c:3615-3627) served by the device.
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
 * (INTEGRATION.md A4); oracle/Makefile builds it into oracle/_ref/libhts_gpu.so, where the reference's cram_get_bam_seq / cram_put_bam_seq / cram_seek /
 * cram_flush / cram_close are renamed hg_ref_* (objcopy --redefine-sym) and the functions below take their names.  libhtsgpu.so and libhts_bgzf.so do not contain it.
 * (The write direction -- cram_put_bam_seq -- has its own banner further down.)
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
    if (fd->mode == 'w') return NULL;
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

/* one record in bam_write1's layout into a bam1_t: the memory layout bam_read1 builds (sam.c:808-866; QNAME padded with NULs to a multiple of four).
 * Returns l_data, or -1 for a malformed record; *used = bytes of the stream the record took. */
static int record_to_bam1(const uint8_t *p, uint64_t avail, bam1_t *b, uint64_t *used) {
    if (avail < 36) return -1;
    const uint32_t block_len = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    if (block_len < 32 || (uint64_t)block_len + 4 > avail) return -1;
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
    *used = (uint64_t)block_len + 4;
    return (int)l_data;
}
static int hand_out(reader *R, bam1_t *b) {
    uint64_t used = 0;
    const int n = record_to_bam1(R->cur->bam + R->pos, R->cur->bam_len - R->pos, b, &used);
    if (n >= 0) R->pos += used;
    return n;
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

/* =====================================================================================================================================================
 * The write direction: cram_put_bam_seq (reference cram/cram_encode.c:4042-4200: records gathered into a container, cram_encode_container ->
 * cram_encode_slice + cram_compress_slice per slice on the pool, cram_flush_container).  Here the records are gathered -- as bam_write1 would lay them out --
 * into RUNS of a few hundred slices; a run goes to hg_cram_writer_containers_host (record encoder on the device, every series block of every slice through
 * the method auto-tuner in a few batched rounds, container framing + CRCs) on a helper thread while the caller fills the next run, and the containers
 * are written to the file in order.  File definition, SAM header container (cram_write_SAM_hdr at open) and EOF container (cram_close) stay the
 * reference's.  A valid CRAM is not a unique byte string: these containers hold one slice each, every series in an EXTERNAL block of its own (what
 * htslib writes for sorted data), methods chosen by the same trial scheme over the same method sets -- stock htslib reads them (tests/test_libhts_gpu.py).
 *
 * The reference's writer runs instead when the cram_fd asks for something this writer does not do -- CRAM 2.x / 4.x, several slices per container, embedded or
 * no reference, lossy names, forced AP delta, bzip2 / lzma / fqzcomp method sets, multi_seq_per_slice forced on or off, bases_per_slice changed, an index
 * written on the fly -- decided at the first record; and from the first run on that meets something it cannot encode (a missing reference, a CIGAR of more
 * than 65 535 operations, a record the device encoder declines): that run and everything after it is REPLAYED through the reference's cram_put_bam_seq in order.
 * ===================================================================================================================================================== */
int hg_ref_cram_put_bam_seq(cram_fd *fd, bam_seq_t *b);
int hg_ref_cram_flush(cram_fd *fd);

typedef struct wrun { uint8_t *recs; size_t len, cap; uint64_t nrec, bases; int32_t *tids; int ntid, tid_cap; int hard; } wrun;   /* hard: holds a record only the reference can write */
typedef struct writer {
    struct writer *next;
    cram_fd *fd; hg_ctx *ctx; hg_cram_writer *W;
    volatile int pass;                             /* the reference's writer owns the cram_fd from here on */
    volatile int failed;                           /* an I/O or replay error: reported by the next call */
    wrun r[2]; int fill;                           /* the run being filled by the caller; the other one may be with the helper */
    pthread_t th; int th_on; pthread_mutex_t m; pthread_cond_t cv; volatile int busy; int job, stop;   /* job: index of the run handed to the helper, -1 none */
    int nrg; char **rg_names;
    uint8_t *out; size_t out_cap;
    uint64_t want, runs, tot_rec; double t_enc, t_wait; int stats;
} writer;
static writer *g_writers;

static __thread cram_fd *tl_wfd; static __thread writer *tl_wr; static __thread unsigned tl_wgen;      /* the writers' per-thread cache (as tl_fd / tl_rd for readers) */
static writer *find_writer(cram_fd *fd) {
    if (tl_wfd == fd && tl_wgen == g_gen) return tl_wr;
    pthread_mutex_lock(&g_lock);
    writer *w = g_writers;
    while (w && w->fd != fd) w = w->next;
    tl_wfd = fd; tl_wr = w; tl_wgen = g_gen;
    pthread_mutex_unlock(&g_lock);
    return w;
}

/* the records of a run through the reference's own writer, in order (the fall-back) */
static int replay_run(writer *Wt, wrun *u) {
    bam1_t *b = bam_init1();
    if (!b) return -1;
    int rc = 0;
    for (uint64_t pos = 0; pos < u->len && rc == 0;) {
        uint64_t used = 0;
        if (record_to_bam1(u->recs + pos, u->len - pos, b, &used) < 0) { rc = -1; break; }
        pos += used;
        if (hg_ref_cram_put_bam_seq(Wt->fd, b) < 0) rc = -1;
    }
    bam_destroy1(b);
    u->len = 0; u->nrec = 0; u->bases = 0; u->ntid = 0; u->hard = 0;
    return rc;
}

/* one run on the device -> containers -> the file.  Returns 0 done, 1 "the reference's writer has to take this run", -1 I/O error */
static int encode_run(writer *Wt, wrun *u) {
    cram_fd *fd = Wt->fd;
    if (u->hard) return 1;
    const double t0 = now_s();
    const int nref = sam_hdr_nref(fd->header);
    hg_cram_ref_seq *refs = calloc((size_t)(nref > 0 ? nref : 1), sizeof *refs);
    if (!refs) return 1;
    int held = 0, ok = 1;
    for (int i = 0; i < u->ntid && ok; i++) {                               /* the reference sequences of this run, whole (cram_get_ref pins them: cram_io.c:3409) */
        const int id = u->tids[i];
        if (id < 0) continue;
        if (id >= nref || !fd->refs || id >= fd->refs->nref) { ok = 0; break; }
        char *seq = cram_get_ref(fd, id, 1, 0);
        if (!seq) { ok = 0; break; }
        held = i + 1;
        pthread_mutex_lock(&fd->refs->lock);
        refs[id].len = (uint64_t)fd->refs->ref_id[id]->length;
        pthread_mutex_unlock(&fd->refs->lock);
        refs[id].bases = (const uint8_t *)seq;
    }
    int rc = 1;
    if (ok) {
        for (int attempt = 0; attempt < 2; attempt++) {
            const size_t need = attempt == 0 ? u->len + u->len / 2 + ((size_t)4 << 20) : Wt->out_cap;
            if (big_grow(&Wt->out, &Wt->out_cap, 0, need) < 0) break;
            uint64_t bytes = 0, nrec = 0;
            const int e = hg_cram_writer_containers_host(Wt->ctx, Wt->W, u->recs, u->len, refs, nref, (const char *const *)Wt->rg_names, Wt->nrg, Wt->out, Wt->out_cap, &bytes, &nrec);
            if (e == HG_ENOMEM && attempt == 0 && bytes > Wt->out_cap) { if (big_grow(&Wt->out, &Wt->out_cap, 0, (size_t)bytes + ((size_t)1 << 20)) < 0) break; continue; }
            if (e == HG_OK && nrec == u->nrec) rc = hwrite(fd->fp, Wt->out, (size_t)bytes) == (ssize_t)bytes ? 0 : -1;
            break;
        }
    }
    for (int i = 0; i < held; i++) if (u->tids[i] >= 0) cram_ref_decr(fd->refs, u->tids[i]);
    free(refs);
    if (rc == 0) {
        fd->record_counter += (int64_t)u->nrec;                             /* (container numbering continues from here if the reference's writer ever takes over) */
        Wt->tot_rec += u->nrec; Wt->runs++; Wt->t_enc += now_s() - t0;
        u->len = 0; u->nrec = 0; u->bases = 0; u->ntid = 0; u->hard = 0;
    }
    return rc;
}

static void *writer_helper(void *arg) {
    writer *Wt = (writer *)arg;
    pthread_mutex_lock(&Wt->m);
    for (;;) {
        while (Wt->job < 0 && !Wt->stop) pthread_cond_wait(&Wt->cv, &Wt->m);
        if (Wt->job < 0) break;
        wrun *u = &Wt->r[Wt->job];
        pthread_mutex_unlock(&Wt->m);
        int rc = Wt->pass ? 1 : encode_run(Wt, u);
        if (rc == 1) { Wt->pass = 1; rc = replay_run(Wt, u); }                 /* in order: nothing later has been written yet */
        pthread_mutex_lock(&Wt->m);
        if (rc < 0) Wt->failed = 1;
        Wt->job = -1; Wt->busy = 0;
        pthread_cond_broadcast(&Wt->cv);
    }
    pthread_mutex_unlock(&Wt->m);
    return NULL;
}

static void writer_wait_idle(writer *Wt) {
    const double t0 = now_s();
    pthread_mutex_lock(&Wt->m);
    while (Wt->busy) pthread_cond_wait(&Wt->cv, &Wt->m);
    pthread_mutex_unlock(&Wt->m);
    Wt->t_wait += now_s() - t0;
}

/* hand the filled run to the helper (waiting for the previous one first: runs are written in order) and start filling the other buffer */
static int writer_submit(writer *Wt, int wait_done) {
    wrun *u = &Wt->r[Wt->fill];
    if (u->nrec) {
        writer_wait_idle(Wt);
        pthread_mutex_lock(&Wt->m);
        Wt->job = Wt->fill; Wt->busy = 1;
        pthread_cond_broadcast(&Wt->cv);
        pthread_mutex_unlock(&Wt->m);
        Wt->fill ^= 1;
        Wt->want = (uint64_t)1024 * (uint64_t)(Wt->fd->seqs_per_slice > 0 ? Wt->fd->seqs_per_slice : 10000);
    }
    if (wait_done) writer_wait_idle(Wt);
    return Wt->failed ? -1 : 0;
}

static int writer_eligible(cram_fd *fd) {
    static int enabled = -1;
    if (enabled < 0) { const char *e = getenv("HTS_GPU_CRAM_SLICE"); enabled = !(e && e[0] == '0'); }
    if (!enabled || !fd || fd->mode != 'w' || !fd->fp || !fd->header || fd->ctr || fd->ctr_mt) return 0;
    if (CRAM_MAJOR_VERS(fd->version) != 3 || CRAM_MINOR_VERS(fd->version) > 1 || fd->level < 1) return 0;       /* (level 0 = every block RAW: the reference's writer) */
    if (fd->slices_per_container != 1 || fd->seqs_per_slice < 1 || fd->bases_per_slice != fd->seqs_per_slice * 500) return 0;
    if (fd->embed_ref > 0 || fd->no_ref || fd->lossy_read_names || fd->ap_delta || fd->multi_seq_user != -1 || fd->idxfp) return 0;
    if (fd->use_bz2 || fd->use_lzma || fd->use_fqz || !fd->use_rans) return 0;
    if (CRAM_MINOR_VERS(fd->version) == 1 ? !fd->use_tok : fd->use_arith) return 0;
    return 1;
}

static void free_writer(writer *Wt) {
    if (Wt->th_on) {
        pthread_mutex_lock(&Wt->m); Wt->stop = 1; pthread_cond_broadcast(&Wt->cv); pthread_mutex_unlock(&Wt->m);
        pthread_join(Wt->th, NULL);
    }
    if (Wt->stats)
        fprintf(stderr, "[htsgpu stats] cram writer: %llu runs, %llu records through the device%s; helper busy %.3f s, caller waited for it %.3f s\n", (unsigned long long)Wt->runs,
                (unsigned long long)Wt->tot_rec, Wt->pass ? ", then the reference's writer" : "", Wt->t_enc, Wt->t_wait);
    for (int i = 0; i < 2; i++) { free(Wt->r[i].recs); free(Wt->r[i].tids); }
    for (int i = 0; i < Wt->nrg; i++) free(Wt->rg_names[i]);
    free(Wt->rg_names); free(Wt->out);
    hg_cram_writer_free(Wt->W);
    pthread_mutex_destroy(&Wt->m); pthread_cond_destroy(&Wt->cv);
    free(Wt);
}

static writer *start_writer(cram_fd *fd) {
    hg_ctx *ctx = hg_front_shared_engine();
    if (!ctx) return NULL;
    writer *Wt = calloc(1, sizeof *Wt);
    if (!Wt) return NULL;
    Wt->fd = fd; Wt->ctx = ctx; Wt->job = -1;
    const int v31 = CRAM_MINOR_VERS(fd->version) == 1;
    Wt->W = hg_cram_writer_new((uint32_t)fd->seqs_per_slice, fd->level, v31 ? (HG_CRAM_WRITE_V31 | (fd->use_arith ? HG_CRAM_WRITE_ARITH : 0)) : 0);
    Wt->nrg = sam_hdr_count_lines(fd->header, "RG");
    if (Wt->nrg < 0) Wt->nrg = 0;
    Wt->rg_names = calloc((size_t)Wt->nrg + 1, sizeof *Wt->rg_names);
    int ok = Wt->W && Wt->rg_names;
    for (int i = 0; ok && i < Wt->nrg; i++) {
        const char *id = sam_hdr_line_name(fd->header, "RG", i);
        if (!id || !(Wt->rg_names[i] = strdup(id))) ok = 0;
    }
    Wt->want = (uint64_t)256 * (uint64_t)fd->seqs_per_slice;                /* the first run; 1024 slices from then on (a run costs about the same whatever it holds) */
    pthread_mutex_init(&Wt->m, NULL); pthread_cond_init(&Wt->cv, NULL);
    { const char *e = getenv("HTS_GPU_STATS"); Wt->stats = e && e[0] == '1'; }
    if (!ok || pthread_create(&Wt->th, NULL, writer_helper, Wt) != 0) { free_writer(Wt); return NULL; }
    Wt->th_on = 1;
    pthread_mutex_lock(&g_lock);
    Wt->next = g_writers; g_writers = Wt; g_gen++;
    pthread_mutex_unlock(&g_lock);
    return Wt;
}

static writer *detach_writer(cram_fd *fd) {
    pthread_mutex_lock(&g_lock);
    writer **pp = &g_writers, *Wt = NULL;
    while (*pp && (*pp)->fd != fd) pp = &(*pp)->next;
    if (*pp) { Wt = *pp; *pp = Wt->next; g_gen++; }
    pthread_mutex_unlock(&g_lock);
    return Wt;
}

int cram_put_bam_seq(cram_fd *fd, bam_seq_t *b) {
    writer *Wt = fd && fd->mode == 'w' ? find_writer(fd) : NULL;
    if (!Wt) {
        if (!writer_eligible(fd) || !(Wt = start_writer(fd))) return hg_ref_cram_put_bam_seq(fd, b);
        tl_wfd = fd; tl_wr = Wt; tl_wgen = g_gen;
    }
    if (Wt->failed) return -1;
    if (Wt->pass) {
        if (Wt->busy || Wt->r[Wt->fill].nrec) {                              /* records still with us go first */
            if (writer_submit(Wt, 1) < 0) return -1;
        }
        return hg_ref_cram_put_bam_seq(fd, b);
    }
    /* the record as bam_write1 lays it out (sam.c:868-928): block_size, the 32 fixed bytes, QNAME without its padding, the rest */
    wrun *u = &Wt->r[Wt->fill];
    const bam1_core_t *c = &b->core;
    const uint32_t l_name = (uint32_t)c->l_qname - c->l_extranul, body = (uint32_t)b->l_data - c->l_extranul, block_len = body + 32;
    if (u->len + block_len + 4 > u->cap) {
        size_t cap = u->cap ? u->cap : (size_t)64 << 20;
        while (cap < u->len + block_len + 4) cap *= 2;
        uint8_t *q = realloc(u->recs, cap);
        if (!q) return -1;
        u->recs = q; u->cap = cap;
    }
    if (c->n_cigar > 0xffff || c->pos > INT32_MAX || c->mpos > INT32_MAX || l_name > 255 || c->tid < -1) u->hard = 1;      /* bam_write1's special cases: the reference's business */
    uint8_t *p = u->recs + u->len;
    #define P32(q, v) do { const uint32_t v_ = (uint32_t)(v); (q)[0] = (uint8_t)v_; (q)[1] = (uint8_t)(v_ >> 8); (q)[2] = (uint8_t)(v_ >> 16); (q)[3] = (uint8_t)(v_ >> 24); } while (0)
    P32(p, block_len); P32(p + 4, c->tid); P32(p + 8, c->pos);
    P32(p + 12, (uint32_t)c->bin << 16 | (uint32_t)c->qual << 8 | (l_name & 0xff));
    P32(p + 16, (uint32_t)c->flag << 16 | (c->n_cigar & 0xffff));
    P32(p + 20, c->l_qseq); P32(p + 24, c->mtid); P32(p + 28, c->mpos); P32(p + 32, c->isize);
    #undef P32
    memcpy(p + 36, b->data, l_name);
    memcpy(p + 36 + l_name, b->data + c->l_qname, body - l_name);
    u->len += (size_t)block_len + 4; u->nrec++; u->bases += (uint64_t)(c->l_qseq > 0 ? c->l_qseq : 0);
    if (u->ntid == 0 || u->tids[u->ntid - 1] != c->tid) {                    /* the references this run touches (sorted input: a handful; unsorted: deduplicated below) */
        int seen = 0;
        for (int i = 0; i < u->ntid && !seen; i++) seen = u->tids[i] == c->tid;
        if (!seen) {
            if (u->ntid == u->tid_cap) { const int cap = u->tid_cap ? 2 * u->tid_cap : 64; int32_t *t = realloc(u->tids, (size_t)cap * sizeof *t); if (!t) return -1; u->tids = t; u->tid_cap = cap; }
            u->tids[u->ntid++] = c->tid;
        }
    }
    if (u->bases > u->nrec * 500 + 1000000) u->hard = 1;                    /* long reads: the reference cuts its slices by bases (bases_per_slice), this writer by records */
    if (u->nrec >= Wt->want || u->len >= ((size_t)3 << 29)) return writer_submit(Wt, 0);
    return 0;
}

int cram_flush(cram_fd *fd) {
    writer *Wt = fd && fd->mode == 'w' ? find_writer(fd) : NULL;
    if (Wt && writer_submit(Wt, 1) < 0) return -1;
    return hg_ref_cram_flush(fd);
}

int cram_close(cram_fd *fd) {
    reader *R = fd && fd->mode != 'w' ? detach_reader(fd) : NULL;
    if (R) free_reader(R);
    int bad = 0;
    writer *Wt = fd && fd->mode == 'w' ? detach_writer(fd) : NULL;
    if (Wt) { bad = writer_submit(Wt, 1) < 0; free_writer(Wt); }
    const int rc = hg_ref_cram_close(fd);
    return bad ? -1 : rc;
}
