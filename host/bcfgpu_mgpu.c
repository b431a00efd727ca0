/*  bcfgpu_mgpu.c -- the pipeline of bcfgpu_host.c over several GPUs of one node, in plain C over the C-ABI: a region of
 *  pileup columns is cut into contiguous shards (the reference's own multi-region mechanism, mpileup.c:652-683), one
 *  host thread and one context per GPU run glfgen+errmod -> combine -> call -m on their shard with no exchange on the data
 *  path, the records that will be written are compacted on each device (bcfgpu_compact_calls) and gathered to rank 0 in
 *  rank order -- which is genomic order, what `bcftools concat` does with per-region files (vcfconcat.c:420) -- and rank 0
 *  walks the gathered buffer and prints the records.
 *
 *      bcfgpu_mgpu <n_sites> <n_smpl> <depth> <seed> [-v] [--gpus N] [--gather rccl|host] [--share-devices]
 *
 *  --gather rccl (default for N > 1): ncclSend / ncclRecv over xGMI (bcfgpu_gather_bytes).
 *  --gather host: every rank copies its buffer to host memory and rank 0 concatenates (no device-to-device link needed).
 *  --share-devices: rank r runs on device r % (devices present) -- a rehearsal of the N-rank code path on fewer GPUs
 *                   (RCCL wants one device per rank, so this implies --gather host).
 *  The output is that of bcfgpu_host for the same arguments, whatever N is (tests/test_c_host.py).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include "bcfgpu.h"

#define READ_LEN 100
static uint64_t rng_state;
static uint32_t rnd32(void)
{
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 2685821657736338717ULL) >> 32);
}
static uint32_t rnd_below(uint32_t n) { return (uint32_t)(((uint64_t)rnd32() * n) >> 32); }

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s (%d)\n", #call, bcfgpu_last_error(), rc_); exit(1); } } while (0)

static void *dev_alloc(bcfgpu_ctx *ctx, size_t bytes)
{
    void *p = NULL;
    CHECK(bcfgpu_malloc(ctx, bytes ? bytes : 16, &p));
    CHECK(bcfgpu_memset(ctx, p, 0, bytes ? bytes : 16));
    return p;
}

/* the whole region on the host (the stand-in for the BAM readers), shared read-only by the ranks */
static int n_sites, n_smpl, varonly, n_ranks, use_rccl;
static int8_t *ref16; static uint32_t *off, *rd; static uint8_t *epos;
static bcfgpu_ctx **ctxs; static bcfgpu_comm *comm;
static uint64_t *counts;                      /* bytes every rank contributes */
static unsigned char **h_part;                /* --gather host: the ranks' buffers in host memory */
static void *d_all; static unsigned char *h_all;
static pthread_barrier_t bar;

static void *rank_main(void *arg)
{
    const int r = (int)(intptr_t)arg;
    bcfgpu_ctx *ctx = ctxs[r];
    const int s0 = (int)((long)n_sites * r / n_ranks), s1 = (int)((long)n_sites * (r + 1) / n_ranks), ns = s1 - s0;
    const size_t c0 = (size_t)s0 * n_smpl, ncell = (size_t)ns * n_smpl;
    const uint32_t r0 = off[c0], nr = off[c0 + ncell] - r0;
    void *d_buf = NULL; uint64_t nbytes = 0; uint32_t nrec = 0;
    if (ns) {
        /* the shard's tile: offsets rebased to its first read */
        uint32_t *loff = malloc((ncell + 1) * sizeof *loff);
        for (size_t i = 0; i <= ncell; ++i) loff[i] = off[c0 + i] - r0;
        int8_t *d_ref = dev_alloc(ctx, (size_t)ns);
        uint32_t *d_off = dev_alloc(ctx, (ncell + 1) * 4), *d_rd = dev_alloc(ctx, ((size_t)nr + 4) * 4);
        uint8_t *d_ep = dev_alloc(ctx, (size_t)nr + 16);
        CHECK(bcfgpu_memcpy_h2d(ctx, d_ref, ref16 + s0, (size_t)ns));
        CHECK(bcfgpu_memcpy_h2d(ctx, d_off, loff, (ncell + 1) * 4));
        CHECK(bcfgpu_memcpy_h2d(ctx, d_rd, rd + r0, (size_t)nr * 4));
        CHECK(bcfgpu_memcpy_h2d(ctx, d_ep, epos + r0, (size_t)nr));
        free(loff);
        bcfgpu_mplp_out mo; memset(&mo, 0, sizeof mo);
        mo.site = dev_alloc(ctx, (size_t)ns * sizeof(bcfgpu_site)); mo.pl = dev_alloc(ctx, ncell * BCFGPU_MAX_PL); mo.dp4 = dev_alloc(ctx, ncell * 4 * sizeof(uint16_t));
        bcfgpu_call_out co; memset(&co, 0, sizeof co);
        co.site = dev_alloc(ctx, (size_t)ns * sizeof(bcfgpu_call_site)); co.gt = dev_alloc(ctx, ncell * 2);
        co.pl = dev_alloc(ctx, ncell * BCFGPU_MAX_PL * sizeof(int32_t));
        bcfgpu_tile tile; memset(&tile, 0, sizeof tile);
        tile.n_sites = ns; tile.n_reads = nr; tile.ref16 = d_ref; tile.plp_off = d_off; tile.rd = d_rd; tile.epos = d_ep;
        CHECK(bcfgpu_pipeline(ctx, &tile, NULL, NULL, &mo, &co));
        /* the records to write, packed on the device; the buffer is sized for every site being one */
        const uint64_t cap = (uint64_t)ns * (sizeof(bcfgpu_call_rec) + 16 + ((2 * (size_t)n_smpl + 3) & ~(size_t)3) + (size_t)BCFGPU_MAX_PL * 4 * n_smpl);
        d_buf = dev_alloc(ctx, cap);
        CHECK(bcfgpu_compact_calls(ctx, ns, s0, mo.site, &co, BCFGPU_MAX_PL, varonly, d_buf, cap, &nbytes, &nrec));
        CHECK(bcfgpu_sync(ctx));
        bcfgpu_free(ctx, d_ref); bcfgpu_free(ctx, d_off); bcfgpu_free(ctx, d_rd); bcfgpu_free(ctx, d_ep);
        bcfgpu_free(ctx, mo.site); bcfgpu_free(ctx, mo.pl); bcfgpu_free(ctx, mo.dp4);
        bcfgpu_free(ctx, co.site); bcfgpu_free(ctx, co.gt); bcfgpu_free(ctx, co.pl);
    }
    counts[r] = nbytes;
    pthread_barrier_wait(&bar);               /* every rank's byte count is known to all */
    uint64_t total = 0;
    for (int i = 0; i < n_ranks; ++i) total += counts[i];
    if (use_rccl) {
        if (r == 0) d_all = dev_alloc(ctx, total);
        CHECK(bcfgpu_gather_bytes(comm, r, d_buf, counts, r == 0 ? d_all : NULL));
        CHECK(bcfgpu_sync(ctx));
        if (r == 0) { h_all = malloc(total ? total : 1); CHECK(bcfgpu_memcpy_d2h(ctx, h_all, d_all, total)); }
    } else {
        h_part[r] = malloc(nbytes ? nbytes : 1);
        if (nbytes) CHECK(bcfgpu_memcpy_d2h(ctx, h_part[r], d_buf, nbytes));
    }
    if (d_buf) bcfgpu_free(ctx, d_buf);
    pthread_barrier_wait(&bar);
    (void)nrec;
    return NULL;
}

/* what bcfgpu_host prints for one record */
static void print_rec(const bcfgpu_call_rec *h)
{
    static const char nt[] = "ACGTN";
    const bcfgpu_site *m = &h->mplp; const bcfgpu_call_site *c = &h->call;
    const int8_t *gt = (const int8_t*)(h + 1);
    if (c->ret <= 0) return;
    printf("%d\t%c\t", h->site + 1, nt[m->a[0] < 0 ? 4 : m->a[0]]);
    int first = 1;
    for (int i = 1; i < m->n_alleles; i++) {
        if (c->als_map[i] <= 0) continue;
        printf("%s%c", first ? "" : ",", i == m->unseen ? '*' : nt[m->a[i]]);
        first = 0;
    }
    if (first) printf(".");
    if (c->qual_missing) printf("\t."); else printf("\t%.4g", c->qual);
    printf("\t%d\t", c->an);
    for (int i = 1; i < c->nals_new; i++) printf("%s%d", i > 1 ? "," : "", c->ac[i]);
    if (c->nals_new < 2) printf(".");
    printf("\t%u\t", m->depth);
    if (gt[0] < 0) printf("./.\n"); else printf("%d/%d\n", gt[0], gt[n_smpl]);
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s n_sites n_smpl depth seed [-v] [--gpus N] [--gather rccl|host] [--share-devices]\n", argv[0]); return 2; }
    n_sites = atoi(argv[1]); n_smpl = atoi(argv[2]);
    const int depth = atoi(argv[3]);
    rng_state = strtoull(argv[4], NULL, 10) * 2 + 1;
    int share = 0; const char *gather = NULL;
    n_ranks = 1;
    for (int i = 5; i < argc; ++i) {
        if (!strcmp(argv[i], "-v")) varonly = 1;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) n_ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gather") && i + 1 < argc) gather = argv[++i];
        else if (!strcmp(argv[i], "--share-devices")) share = 1;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    const int ndev = bcfgpu_device_count();
    if (n_ranks < 1) n_ranks = 1;
    if (!share && n_ranks > ndev) { fprintf(stderr, "%d ranks but %d device(s): use --share-devices to rehearse\n", n_ranks, ndev); return 1; }
    use_rccl = n_ranks > 1 && !share && (!gather || !strcmp(gather, "rccl"));
    if (n_ranks > 1)
        fprintf(stderr, "[bcfgpu_mgpu] %d ranks on %d device(s); ordered gather over %s\n", n_ranks, ndev,
                use_rccl ? "RCCL (bcfgpu_gather_bytes: grouped ncclSend / ncclRecv)" : "host memory (--gather host / --share-devices)");
    static const uint8_t bq_values[4] = { 11, 25, 37, 40 };

    /* ---- the column loop of bcfgpu_host.c: the whole region ---- */
    const size_t ncell = (size_t)n_sites * n_smpl, max_reads = ncell * (size_t)(2 * depth);
    ref16 = malloc((size_t)n_sites + 1); off = malloc((ncell + 1) * sizeof *off); rd = malloc((max_reads + 4) * sizeof *rd); epos = malloc(max_reads + 16);
    if (!ref16 || !off || !rd || !epos) { fprintf(stderr, "out of memory\n"); return 1; }
    size_t nr = 0;
    off[0] = 0;
    for (int k = 0; k < n_sites; k++) {
        const int ref2 = (int)rnd_below(4), alt2 = (ref2 + 1 + (int)rnd_below(3)) & 3;
        const int is_var = rnd_below(4) == 0;
        ref16[k] = (int8_t)(1 << ref2);
        for (int s = 0; s < n_smpl; s++) {
            const int nalt = is_var ? (int)rnd_below(3) : 0;
            const int n = depth + (int)rnd_below((uint32_t)depth);
            for (int j = 0; j < n; j++) {
                const int bq = bq_values[rnd_below(4)];
                int base = (nalt == 2 || (nalt == 1 && (rnd32() & 1))) ? alt2 : ref2;
                if (rnd_below(1000) < (bq < 20 ? 80u : 3u)) base = (base + 1 + (int)rnd_below(3)) & 3;
                const int mapq = rnd_below(10) ? 60 : (int)rnd_below(60);
                const int qpos = (int)rnd_below(READ_LEN);
                const uint32_t one_match = (uint32_t)READ_LEN << 4;
                bcfgpu_pack_read(1 << base, bq, mapq, (int)(rnd32() & 1), 0, 0, 0, qpos, READ_LEN, &one_match, 1, 1, &rd[nr], &epos[nr]);
                nr++;
            }
            off[(size_t)k * n_smpl + s + 1] = (uint32_t)nr;
        }
    }

    /* ---- one context per rank, sized for its shard ---- */
    ctxs = calloc((size_t)n_ranks, sizeof *ctxs); counts = calloc((size_t)n_ranks, sizeof *counts); h_part = calloc((size_t)n_ranks, sizeof *h_part);
    for (int r = 0; r < n_ranks; ++r) {
        const int s0 = (int)((long)n_sites * r / n_ranks), s1 = (int)((long)n_sites * (r + 1) / n_ranks);
        bcfgpu_cfg cfg; memset(&cfg, 0, sizeof cfg);
        cfg.device = share ? r % ndev : r; cfg.n_smpl = n_smpl; cfg.max_sites = s1 - s0 > 0 ? s1 - s0 : 1;
        cfg.max_reads = (uint64_t)(off[(size_t)s1 * n_smpl] - off[(size_t)s0 * n_smpl]) + 64;
        cfg.min_baseQ = 13; cfg.capQ = 60; cfg.errmod_theta = 0.; cfg.fmt_flag = BCFGPU_INFO_VDB | BCFGPU_INFO_RPB;
        cfg.call_theta = 1.1e-3; cfg.call_flag = varonly ? BCFGPU_CALL_VARONLY : 0; cfg.n_grp = 1; cfg.ploidy_max = 2;
        CHECK(bcfgpu_create(&cfg, &ctxs[r]));
    }
    if (use_rccl) CHECK(bcfgpu_comm_init_all(ctxs, n_ranks, &comm));
    pthread_barrier_init(&bar, NULL, (unsigned)n_ranks);
    pthread_t *th = malloc((size_t)n_ranks * sizeof *th);
    for (int r = 1; r < n_ranks; ++r) if (pthread_create(&th[r], NULL, rank_main, (void*)(intptr_t)r)) { fprintf(stderr, "pthread_create failed\n"); return 1; }
    rank_main((void*)(intptr_t)0);
    for (int r = 1; r < n_ranks; ++r) pthread_join(th[r], NULL);

    /* ---- rank 0: the gathered records, in rank order = genomic order ---- */
    uint64_t nrec = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const unsigned char *p = use_rccl ? NULL : h_part[r];
        static uint64_t base = 0;
        if (use_rccl) p = h_all + base;
        for (uint64_t o = 0; o < counts[r]; ) {
            const bcfgpu_call_rec *h = (const bcfgpu_call_rec*)(p + o);
            print_rec(h);
            o += h->bytes; ++nrec;
        }
        base += counts[r];
    }
    fprintf(stderr, "%d sites, %d samples, %zu reads, %d rank(s), gather %s, %llu records\n", n_sites, n_smpl, nr, n_ranks,
            use_rccl ? "rccl" : "host", (unsigned long long)nrec);
    if (comm) bcfgpu_comm_destroy(comm);
    for (int r = 0; r < n_ranks; ++r) bcfgpu_destroy(ctxs[r]);
    return 0;
}
