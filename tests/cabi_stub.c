/* C twin of the cgo file regengo_amd/codegen.py emits (tests/golden/codegen/, the _gpu.go files): the SAME call sequence against
 * include/rgx.h, function for function, so that a compiler checks the types and a GPU run checks the protocol -- the stand-in
 * for `go build` in an image without a Go toolchain.  Test infrastructure, not product.
 *
 *   <name>Init          blob -> rgx_program_from_blob -> rgx_program_to_device(devices[0]) (+ rgx_sharded_create for a device list)
 *   <name>GetCtx/PutCtx a pool of rgx_stream_ctx, destroyed at exit (the Go file: sync.Pool + finalizer)
 *   FindAllBytesAppend  rgx_find_all_bytes with the RGX_E_CAPACITY retry, rgx_sharded_find_all_bytes when sharded
 *   FindReader          the read loop of streaming.go:110-250 around rgx_find_chunk; a negative status = "this chunk goes to the
 *                       Go loop": the twin has no Go loop, it reports GOFALLBACK <status> and carries on as the stub would
 *   FindReaderCount     the same loop around rgx_count_chunk
 *   MatchBytes / FindBytes / ReplaceAllBytes   rgx_match_bytes / rgx_find_bytes / rgx_replace_all_bytes with its capacity loop
 *
 * usage: cabi_stub <tables.bin> <input file> <command> [args]      (results as text lines on stdout; exit 0 unless the plumbing broke)
 *   info | match | find | findall <n> | reader <bufsize> <maxleftover> <readsize> | count <bufsize> <maxleftover> <readsize> |
 *   runs <bufsize> <maxleftover> <readsize> <blockbytes> [ndev] | countruns ... (FindReader over runs of chunks: rgx_find_chunks) |
 *   replace <template> <first_only> | sharded <ndev> <n>
 * Built with:  gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude tests/cabi_stub.c -Lregengo_amd/lib -lrgx_hip */
#include <rgx.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GPU_THRESHOLD 0 /* the emitted file keeps small inputs on the Go path; the twin sends everything down */

static rgx_program* g_prog;
static rgx_sharded* g_sharded;
static rgx_info g_info;
static int g_ncap;

/* ---- <name>CtxPool ------------------------------------------------------------------------------------------------------ */
#define POOL 4
static rgx_stream_ctx* g_pool[POOL];
static int g_pooled;
static rgx_stream_ctx* get_ctx(void) {
  rgx_stream_ctx* c = NULL;
  if (g_pooled > 0) return g_pool[--g_pooled];
  if (rgx_stream_ctx_create(g_prog, &c) != RGX_OK) return NULL;
  return c;
}
static void put_ctx(rgx_stream_ctx* c) {
  if (g_pooled < POOL) g_pool[g_pooled++] = c;
  else rgx_stream_ctx_destroy(c);
}
static void gpu_freeze(void) { /* <Name>GPUFreeze: the end of what the library learns about this pattern's texts */
  int i;
  if (!g_prog) return;
  rgx_program_freeze(g_prog);
  if (g_sharded) {
    rgx_sharded_info si;
    if (rgx_sharded_shape(g_sharded, &si) == RGX_OK)
      for (i = 0; i < si.n_local; i++) rgx_program_freeze((rgx_program*)rgx_sharded_program(g_sharded, i));
  }
}
static void gpu_close(void) { /* <Name>GPUClose */
  while (g_pooled > 0) rgx_stream_ctx_destroy(g_pool[--g_pooled]);
  if (g_sharded) { rgx_sharded_destroy(g_sharded); g_sharded = NULL; }
  if (g_prog) { rgx_program_destroy(g_prog); g_prog = NULL; }
}

/* ---- <name>Init ------------------------------------------------------------------------------------------------------------ */
static int init(const void* blob, size_t blob_len, const int* devices, int ndev) {
  rgx_program* p = NULL;
  int rc;
  if (rgx_abi_version() != RGX_ABI_VERSION || ndev < 1) return RGX_E_INVALID;
  if ((rc = rgx_program_from_blob(blob, blob_len, &p)) != RGX_OK) return rc;
  if ((rc = rgx_program_to_device(p, devices[0])) != RGX_OK) { rgx_program_destroy(p); return rc; }
  if (ndev > 1) {
    rgx_sharded* sh = NULL;
    if (rgx_sharded_create(blob, blob_len, devices, ndev, &sh) == RGX_OK) g_sharded = sh; /* failure: one device is still right */
  }
  g_prog = p;
  if ((rc = rgx_program_info(p, &g_info)) != RGX_OK) return rc;
  g_ncap = g_info.ncap;
  return RGX_OK;
}

static void print_row(const char* tag, const int32_t* c) {
  int k;
  printf("%s", tag);
  for (k = 0; k < g_ncap; k++) printf(" %d", (int)c[k]);
  printf("\n");
}

/* ---- FindAllBytesAppend ------------------------------------------------------------------------------------------------------ */
static long long find_all(const uint8_t* input, size_t len, long long n, int32_t** out) {
  rgx_stream_ctx* ctx;
  size_t cap, bound;
  int32_t* spans = NULL;
  long long w;
  int minlen = g_info.min_match_len > 1 ? g_info.min_match_len : 1;
  *out = NULL;
  if (n == 0) return 0;
  if ((ctx = get_ctx()) == NULL) return RGX_E_NO_DEVICE;
  cap = len / 64 + 1024;
  bound = len / (size_t)minlen + 1;
  if (cap > bound) cap = bound;
  if (n > 0 && (size_t)n < cap) cap = (size_t)n;
  /* <name>Template of the emitted file: a fixed-template pattern fetches one int32 per match and rebuilds the records */
  int32_t tmpl[64];
  int use_starts = !g_sharded && g_info.fixed_captures && g_ncap <= 64 && rgx_program_capture_template(g_prog, tmpl) >= 0;
  for (;;) {
    rgx_result res;
    free(spans);
    spans = (int32_t*)malloc(cap * (size_t)g_ncap * sizeof(int32_t) + 16);
    if (use_starts) {
      int32_t* starts = (int32_t*)malloc(cap * sizeof(int32_t) + 16);
      long long i;
      int k;
      w = rgx_find_all_starts(g_prog, ctx, input, len, n, starts, cap, &res);
      if (w == RGX_E_UNSUPPORTED) { free(starts); use_starts = 0; continue; }
      if (w >= 0) {
        printf("STARTS %lld\n", w);
        for (i = 0; i < w; i++) for (k = 0; k < g_ncap; k++) spans[i * g_ncap + k] = starts[i] + tmpl[k];
      }
      free(starts);
    } else if (g_sharded) w = rgx_sharded_find_all_bytes(g_sharded, input, len, n, spans, cap, &res);
    else w = rgx_find_all_bytes(g_prog, ctx, input, len, n, spans, cap, &res);
    if (w == RGX_E_CAPACITY && (size_t)res.total > cap) { /* BEFORE the generic fallback: RGX_E_CAPACITY is negative too */
      cap = (size_t)res.total;
      printf("RETRY %lld\n", (long long)res.total);
      continue;
    }
    break;
  }
  put_ctx(ctx);
  if (w < 0) { free(spans); return w; } /* the Go path */
  *out = spans;
  return w;
}

/* ---- the read loop of streaming.go:110-250, shared by FindReader and FindReaderCount -------------------------------------------- */
typedef int (*chunk_fn)(rgx_stream_ctx* ctx, uint8_t* buf, size_t data_len, int is_full, long long stream_offset, int chunk_index,
                        long long max_leftover, void* user, long long* keep);
static int read_loop(FILE* rd, size_t read_size, rgx_stream_config cfg, chunk_fn fn, void* user) {
  rgx_stream_ctx* ctx = get_ctx();
  uint8_t* buf;
  size_t leftover = 0;
  long long stream_offset = 0;
  int chunk_index = 0, rc = 0;
  if (!ctx) return RGX_E_NO_DEVICE;
  buf = (uint8_t*)malloc((size_t)cfg.buffer_size);
  for (;;) {
    size_t want = (size_t)cfg.buffer_size - leftover, n, data_len;
    int eof, is_full;
    long long keep = 0;
    if (read_size && want > read_size) want = read_size; /* a reader that returns short reads */
    n = fread(buf + leftover, 1, want, rd);
    eof = n < want && feof(rd);
    data_len = leftover + n;
    if (data_len == 0) {
      if (eof) break;
      chunk_index++;
      continue;
    }
    is_full = n == (size_t)cfg.buffer_size - leftover;
    rc = fn(ctx, buf, data_len, is_full, stream_offset, chunk_index, cfg.max_leftover, user, &keep);
    if (rc != 0) break; /* stop requested */
    if (is_full) {
      leftover = data_len - (size_t)keep;
      stream_offset += keep;
      memmove(buf, buf + keep, leftover);
    } else {
      leftover = 0;
    }
    chunk_index++;
    if (eof) break;
  }
  free(buf);
  put_ctx(ctx);
  return rc < 0 ? rc : 0;
}

/* `held`: the reused result struct of FindReader (streaming.go:117) for a Tagged-DFA program -- per group the slice (lo, hi) of buf it
 * was last assigned; the engine assigns a group only when its start tag is set (tdfa.go:1031-1046; the record says (-1, -1)
 * otherwise), so a field may still alias the bytes of an earlier match's group -- whatever lies there NOW (FIELD lines). */
typedef struct reader_state { int32_t* spans; size_t cap; long long count; int32_t held[128]; int have[64]; } reader_state;
static int reader_chunk(rgx_stream_ctx* ctx, uint8_t* buf, size_t data_len, int is_full, long long stream_offset, int chunk_index,
                        long long max_leftover, void* user, long long* keep) {
  reader_state* st = (reader_state*)user;
  int64_t committed = 0, keep_from = 0;
  long long w = rgx_find_chunk(g_prog, ctx, buf, data_len, is_full, max_leftover, st->spans, st->cap, &committed, &keep_from, NULL);
  long long i;
  if (w < 0) { /* findReaderChunkGo would take this chunk; the twin can only say so (keep = what the Go loop would return) */
    printf("GOFALLBACK %lld chunk %d\n", w, chunk_index);
    *keep = is_full ? (long long)data_len - max_leftover : 0;
    return 0;
  }
  for (i = 0; i < w; i++) {
    const int32_t* c = st->spans + i * g_ncap;
    printf("MATCH %lld %d %.*s\n", stream_offset + c[0], chunk_index, (int)(c[1] - c[0]), (const char*)buf + c[0]);
    if (g_info.ref_find_engine == 1 && !(g_info.flags & RGX_FLAG_STDLIB_SEMANTICS)) {
      int g;
      for (g = 0; g < g_ncap / 2 && g < 64; g++) {
        if (c[2 * g] >= 0) { st->held[2 * g] = c[2 * g]; st->held[2 * g + 1] = c[2 * g + 1]; st->have[g] = 1; }
        if (st->have[g]) printf("FIELD %d %.*s\n", g, (int)(st->held[2 * g + 1] - st->held[2 * g]), (const char*)buf + st->held[2 * g]);
        else printf("FIELD %d <nil>\n", g);
      }
    }
  }
  st->count += w;
  *keep = keep_from;
  return 0;
}
static int count_chunk(rgx_stream_ctx* ctx, uint8_t* buf, size_t data_len, int is_full, long long stream_offset, int chunk_index,
                       long long max_leftover, void* user, long long* keep) {
  reader_state* st = (reader_state*)user;
  int64_t committed = 0, keep_from = 0;
  long long w = rgx_count_chunk(g_prog, ctx, buf, data_len, is_full, max_leftover, &committed, &keep_from, NULL);
  (void)stream_offset;
  if (w < 0) {
    printf("GOFALLBACK %lld chunk %d\n", w, chunk_index);
    *keep = is_full ? (long long)data_len - max_leftover : 0;
    return 0;
  }
  st->count += w;
  *keep = keep_from;
  return 0;
}

static int find_reader(FILE* rd, long long bufsize, long long max_leftover, size_t read_size, int count_only) {
  rgx_stream_config in, cfg;
  reader_state st;
  int rc;
  in.buffer_size = bufsize; in.max_leftover = max_leftover;
  rc = rgx_stream_config_resolve(g_prog, &in, &cfg); /* cfg.Validate + ApplyDefaults */
  if (rc != RGX_OK) { printf("CONFIG_ERROR %d %s\n", rc, rgx_status_str(rc)); return 0; }
  st.cap = (size_t)cfg.buffer_size / (size_t)(g_info.min_match_len > 1 ? g_info.min_match_len : 1) + 1;
  st.spans = (int32_t*)malloc(st.cap * (size_t)g_ncap * sizeof(int32_t) + 16);
  st.count = 0;
  memset(st.have, 0, sizeof st.have);
  rc = read_loop(rd, read_size, cfg, count_only ? count_chunk : reader_chunk, &st);
  printf("COUNT %lld\n", st.count);
  free(st.spans);
  return rc;
}

/* ---- FindReader over RUNS of chunks (round 6; the emitted <name>ReadRuns + FindReader): the reference's reads -- BufferSize bytes, then
 * BufferSize - MaxLeftover at a time -- appended to ONE block while they come back full; a short read, EOF or a full block ends the run,
 * and the run's chunks are answered by one rgx_find_chunks call (several devices: a rgx_sharded_round of chunk ranges + the gather of
 * the rows).  Same MATCH / FIELD lines as `reader`. */
typedef struct run_state { int32_t* spans; size_t cap; long long count; int32_t held[128]; int have[64]; uint8_t* prev; int ndev; int count_only; } run_state;
static long long run_rows(rgx_stream_ctx* ctx, uint8_t* block, size_t fill, rgx_stream_config cfg, int final, int nchunks, run_state* st) {
  const long long B = cfg.buffer_size, ML = cfg.max_leftover, S = B - ML;
  rgx_chunks_result cr;
  long long w;
  if (g_sharded && nchunks >= 2 * st->ndev && !st->count_only) { /* <name>ShardedRun: ranks (here: devices) own chunk ranges */
    rgx_shard_window wins[8];
    rgx_shard_round rounds[8];
    const int per = (nchunks + st->ndev - 1) / st->ndev;
    long long total;
    int i;
    memset(wins, 0, sizeof wins);
    for (i = 0; i < st->ndev; i++) {
      const int k0 = i * per, k1 = k0 + per < nchunks ? k0 + per : nchunks;
      size_t lo, hi;
      if (k0 >= k1) continue;
      lo = (size_t)k0 * (size_t)S;
      hi = (size_t)(k1 - 1) * (size_t)S + (size_t)B;
      if (k1 == nchunks || hi > fill) hi = fill;
      wins[i].buf = block + lo; wins[i].len = hi - lo; wins[i].base = (int64_t)lo; wins[i].is_host = 1;
      wins[i].last = (final && k1 == nchunks) ? 1 : 0;
      wins[i].reader_buffer_size = B; wins[i].reader_max_leftover = ML;
    }
    total = rgx_sharded_round(g_sharded, wins, 0, 0, rounds);
    if (total >= 0) {
      int64_t* rows64 = (int64_t*)malloc(((size_t)total + 1) * (size_t)g_ncap * 8);
      const int64_t n = rgx_sharded_gather(g_sharded, 0, NULL, rows64, (size_t)total, NULL);
      long long j;
      if (n == total) {
        if ((size_t)n > st->cap) { st->cap = (size_t)n; st->spans = (int32_t*)realloc(st->spans, st->cap * (size_t)g_ncap * 4 + 16); }
        for (j = 0; j < n * g_ncap; j++) st->spans[j] = (int32_t)rows64[j];
        free(rows64);
        printf("SHARDEDRUN %d chunks %d\n", st->ndev, nchunks);
        return n;
      }
      free(rows64);
    }
    /* (any failure: one device is still right) */
  }
  for (;;) {
    w = rgx_find_chunks(g_prog, ctx, block, fill, B, ML, final, st->count_only ? NULL : st->spans, st->count_only ? 0 : st->cap, &cr);
    if (w == RGX_E_CAPACITY && (size_t)cr.rows > st->cap) {
      st->cap = (size_t)cr.rows;
      st->spans = (int32_t*)realloc(st->spans, st->cap * (size_t)g_ncap * 4 + 16);
      continue;
    }
    return w;
  }
}
static int find_reader_runs(FILE* rd, long long bufsize, long long max_leftover, size_t read_size, long long block_bytes, int ndev, int count_only) {
  rgx_stream_config in, cfg;
  run_state st;
  rgx_stream_ctx* ctx;
  uint8_t* block;
  long long B, ML, S, nmax, stream_offset = 0;
  size_t leftover = 0;
  int chunk_index = 0, rc, done = 0;
  in.buffer_size = bufsize; in.max_leftover = max_leftover;
  rc = rgx_stream_config_resolve(g_prog, &in, &cfg);
  if (rc != RGX_OK) { printf("CONFIG_ERROR %d %s\n", rc, rgx_status_str(rc)); return 0; }
  B = cfg.buffer_size; ML = cfg.max_leftover; S = B - ML;
  nmax = (block_bytes - B) / S + 1;
  if (nmax < 1) nmax = 1;
  block = (uint8_t*)malloc((size_t)((nmax - 1) * S + B));
  memset(&st, 0, sizeof st);
  st.cap = (size_t)((nmax - 1) * S + B) / 64 + 1024;
  st.spans = (int32_t*)malloc(st.cap * (size_t)g_ncap * 4 + 16);
  st.prev = (uint8_t*)calloc((size_t)B, 1);
  st.ndev = ndev; st.count_only = count_only;
  ctx = get_ctx();
  if (!ctx) return RGX_E_NO_DEVICE;
  while (!done) {
    size_t fill = leftover;
    long long nfull = 0, w, i;
    int final = 0, nchunks;
    while (nfull < nmax) { /* the reference's reads, one per chunk: rd.Read(buf[leftover:]) */
      size_t want = (size_t)(nfull * S + B) - fill, ask = want, n;
      if (read_size && ask > read_size) ask = read_size;
      n = fread(block + fill, 1, ask, rd);
      if (n == 0 && feof(rd)) { final = fill - (size_t)(nfull * S) > 0; done = 1; break; }
      fill += n;
      if (n < want) { final = 1; break; } /* a short read: this chunk is not full (streaming.go:177) */
      nfull++;
    }
    nchunks = (int)nfull + (final ? 1 : 0);
    if (fill > 0 && nchunks > 0) {
      w = run_rows(ctx, block, fill, cfg, final, nchunks, &st);
      if (w < 0) { /* findReaderChunkGo would take the run's chunks one by one; the twin can only say so */
        printf("GOFALLBACK %lld run at chunk %d (%d chunks)\n", w, chunk_index, nchunks);
      } else if (count_only) {
        st.count += w;
      } else {
        for (i = 0; i < w; i++) {
          const int32_t* c = st.spans + i * g_ncap;
          long long k = c[0] / S;
          size_t lo, clen;
          if (k > nchunks - 1) k = nchunks - 1;
          lo = (size_t)(k * S);
          clen = fill - lo < (size_t)B ? fill - lo : (size_t)B;
          printf("MATCH %lld %d %.*s\n", stream_offset + c[0], chunk_index + (int)k, (int)(c[1] - c[0]), (const char*)block + c[0]);
          if (g_info.ref_find_engine == 1 && !(g_info.flags & RGX_FLAG_STDLIB_SEMANTICS)) {
            /* the reused struct's fields are slices of the reference's ONE buffer, which holds chunk k now: a field an earlier match set
             * shows what lies at its offsets in THIS chunk (behind a short chunk's end: what the chunk before left there) */
            const uint8_t* before = lo >= (size_t)S ? block + lo - (size_t)S : st.prev;
            int g;
            for (g = 0; g < g_ncap / 2 && g < 64; g++) {
              if (c[2 * g] >= 0) { st.held[2 * g] = c[2 * g] - (int32_t)lo; st.held[2 * g + 1] = c[2 * g + 1] - (int32_t)lo; st.have[g] = 1; }
              if (st.have[g]) {
                int x;
                printf("FIELD %d ", g);
                for (x = st.held[2 * g]; x < st.held[2 * g + 1]; x++) putchar((size_t)x < clen ? block[lo + (size_t)x] : before[x]);
                putchar('\n');
              } else printf("FIELD %d <nil>\n", g);
            }
          }
        }
        st.count += w;
      }
    }
    if (done) break;
    if (nfull > 0) memcpy(st.prev, block + (size_t)((nfull - 1) * S), (size_t)B);
    stream_offset += nfull * S;
    if (final) { /* behind a short read the reference zeroes leftover and does not advance streamOffset past the short chunk (241-244) */
      memcpy(st.prev, block + (size_t)(nfull * S), fill - (size_t)(nfull * S));
      leftover = 0;
      chunk_index += (int)nfull + 1;
    } else {
      leftover = fill - (size_t)(nfull * S);
      memmove(block, block + (size_t)(nfull * S), leftover);
      chunk_index += (int)nfull;
    }
  }
  printf("COUNT %lld\n", st.count);
  put_ctx(ctx);
  free(block); free(st.spans); free(st.prev);
  return 0;
}

/* ---- <name>Replace ---------------------------------------------------------------------------------------------------------------- */
static int replace_all(const uint8_t* input, size_t len, const char* tmpl, int first_only) {
  rgx_stream_ctx* ctx = get_ctx();
  size_t cap = len + len / 8 + 64;
  uint8_t* out = (uint8_t*)malloc(cap);
  if (!ctx) return RGX_E_NO_DEVICE;
  for (;;) {
    int64_t need = 0;
    long long rc = rgx_replace_all_bytes(g_prog, ctx, input, len, tmpl, strlen(tmpl), first_only, out, cap, &need, NULL);
    if (rc == RGX_E_CAPACITY) {
      free(out);
      cap = (size_t)need + 64;
      out = (uint8_t*)malloc(cap);
      printf("RETRY %lld\n", (long long)need);
      continue;
    }
    if (rc == RGX_E_INVALID) { printf("PANIC invalid replace template: %s\n", rgx_last_error()); break; }
    if (rc < 0) { printf("GOFALLBACK %lld\n", rc); break; }
    printf("OUT %lld\n", rc);
    fwrite(out, 1, (size_t)rc, stdout);
    printf("\nEND\n");
    break;
  }
  free(out);
  put_ctx(ctx);
  return 0;
}

static uint8_t* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  uint8_t* b;
  long n;
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  n = ftell(f);
  fseek(f, 0, SEEK_SET);
  b = (uint8_t*)malloc((size_t)n + 16);
  *len = fread(b, 1, (size_t)n, f);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  size_t blob_len = 0, len = 0;
  uint8_t *blob, *input;
  int devices[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int ndev = 1, rc;
  const char* cmd;
  if (argc < 4) { fprintf(stderr, "usage: cabi_stub <tables.bin> <input> <command> ...\n"); return 2; }
  blob = slurp(argv[1], &blob_len);
  input = slurp(argv[2], &len);
  cmd = argv[3];
  if (!blob || !input) { fprintf(stderr, "cannot read inputs\n"); return 2; }
  if (strcmp(cmd, "sharded") == 0 && argc > 4) ndev = atoi(argv[4]);
  if ((strcmp(cmd, "runs") == 0 || strcmp(cmd, "countruns") == 0) && argc > 8) ndev = atoi(argv[8]);
  rc = init(blob, blob_len, devices, ndev);
  if (strcmp(cmd, "info") == 0) {
    /* works without a device: the blob loads, rgx_program_to_device says RGX_E_NO_DEVICE and the Go path stays */
    rgx_program* p = NULL;
    rgx_info info;
    int rb = rgx_program_from_blob(blob, blob_len, &p);
    printf("INIT %d %s\n", rc, rgx_status_str(rc));
    if (rb == RGX_OK && rgx_program_info(p, &info) == RGX_OK)
      printf("INFO abi %d ncap %d min %d max %d findall %d stream %d find %d match %d engine %d flags %u replace %d\n", info.abi_version, info.ncap,
             info.min_match_len, info.max_match_len, info.ref_findall_offered, info.ref_stream_offered, info.ref_find_offered,
             info.ref_match_offered, info.ref_find_engine, info.flags, info.ref_replace_offered);
    else printf("BLOB_ERROR %d\n", rb);
    if (p) rgx_program_destroy(p);
    gpu_close();
    return 0;
  }
  if (rc != RGX_OK) { printf("INIT %d %s: %s\n", rc, rgx_status_str(rc), rgx_last_error()); return 1; }
  if (getenv("RGX_TWIN_FREEZE")) gpu_freeze(); /* a caller that froze the pattern before its first call: every answer the same */
  if (strcmp(cmd, "match") == 0) {
    rgx_stream_ctx* ctx = get_ctx();
    int m = 0;
    rc = rgx_match_bytes(g_prog, ctx, input, len, &m);
    put_ctx(ctx);
    if (rc != RGX_OK) printf("GOFALLBACK %d\n", rc);
    else printf("MATCHED %d\n", m != 0);
  } else if (strcmp(cmd, "find") == 0) {
    rgx_stream_ctx* ctx = get_ctx();
    int32_t c[64];
    int f = 0;
    rc = rgx_find_bytes(g_prog, ctx, input, len, c, &f);
    put_ctx(ctx);
    if (rc != RGX_OK) printf("GOFALLBACK %d\n", rc);
    else if (!f) printf("NOTFOUND\n");
    else print_row("ROW", c);
  } else if (strcmp(cmd, "findall") == 0 || strcmp(cmd, "sharded") == 0) {
    int32_t* spans = NULL;
    long long n = argc > (strcmp(cmd, "sharded") == 0 ? 5 : 4) ? atoll(argv[strcmp(cmd, "sharded") == 0 ? 5 : 4]) : -1;
    long long w = find_all(input, len, n, &spans), i;
    if (strcmp(cmd, "sharded") == 0) printf("SHARDED %d\n", g_sharded != NULL);
    if (w < 0) printf("GOFALLBACK %lld\n", w);
    else {
      printf("COUNT %lld\n", w);
      for (i = 0; i < w; i++) print_row("ROW", spans + i * g_ncap);
    }
    free(spans);
  } else if (strcmp(cmd, "reader") == 0 || strcmp(cmd, "count") == 0) {
    FILE* rd = fopen(argv[2], "rb");
    if (argc < 7) return 2;
    rc = find_reader(rd, atoll(argv[4]), atoll(argv[5]), (size_t)atoll(argv[6]), strcmp(cmd, "count") == 0);
    fclose(rd);
    if (rc < 0) printf("ERROR %d\n", rc);
  } else if (strcmp(cmd, "runs") == 0 || strcmp(cmd, "countruns") == 0) { /* runs <bufsize> <maxleftover> <readsize> <blockbytes> [ndev] */
    FILE* rd = fopen(argv[2], "rb");
    if (argc < 8) return 2;
    rc = find_reader_runs(rd, atoll(argv[4]), atoll(argv[5]), (size_t)atoll(argv[6]), atoll(argv[7]), ndev, strcmp(cmd, "countruns") == 0);
    fclose(rd);
    if (rc < 0) printf("ERROR %d\n", rc);
  } else if (strcmp(cmd, "replace") == 0) {
    if (argc < 6) return 2;
    replace_all(input, len, argv[4], atoi(argv[5]));
  } else {
    fprintf(stderr, "unknown command %s\n", cmd);
    return 2;
  }
  gpu_close();
  free(blob);
  free(input);
  return 0;
}
