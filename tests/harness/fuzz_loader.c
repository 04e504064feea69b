/*
 * fuzz_loader.c -- the MatrixMarket loader under AddressSanitizer + UBSan (host code only), test program.
 *
 *   fuzz_loader.elf <dir> <iterations> <seed> [archive.mtx.gz archive.mtx.zip ...]
 *
 * Every iteration writes a small or medium coordinate file (general / symmetric, real / integer / pattern),
 * damages it (truncation, byte flips, joined / split / deleted / duplicated lines, odd separators, odd numbers,
 * indices that are 0, negative, out of range or too long, comments and blank lines in the entry section, size lines
 * whose counts wrap around) and reads
 * it twice: through MMRead (the in-memory parser of mmfast.c with its serial fall-back) and through the serial steps
 * alone (banner, size line, bounds pre-pass, MMtoCOO = the fscanf loop that follows the reference's
 * src/lib/parser.c:59-97).  Both must agree: rejected by both, or the same entries in the same order with the same row
 * lengths.  Accepted files also go through MMtoCSR / MMtoELL.  With archives on the command line (.gz / .bz2 / .xz /
 * .zip, valid ones written by the test), every fourth iteration damages one of them instead -- bytes flipped, the file cut
 * short, 32-bit fields set to all ones -- and hands it to extractInTmpFS (gzip / bzip2 / xz decoders and the zip reader
 * of csrc/host/utils.c); what it inflates is then read as a matrix.  The sanitizers abort on any bad access on the way.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "parser.h"
#include "sparseMatrix.h"
#include "utils.h"

static uint64_t rngState;
static uint64_t rnd(void) {                          /* splitmix64 */
    uint64_t z = (rngState += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }

typedef struct { char* s; size_t n, cap; } text;
static void put(text* t, const char* s, size_t n) {
    if (t->n + n + 1 > t->cap) {
        t->cap = (t->n + n + 1) * 2;
        t->s = realloc(t->s, t->cap);
        if (!t->s) { perror("realloc"); exit(2); }
    }
    memcpy(t->s + t->n, s, n);
    t->n += n;
    t->s[t->n] = 0;
}
static void puts_(text* t, const char* s) { put(t, s, strlen(s)); }

static const char* ODD_VALUES[] = {"nan", "inf", "-inf", "infinity", "1e400", "1e-400", "0x1p3", "1.", ".5", "-.5e+3", "+7", "1e", "1e+",
                                   "3.141592653589793238462643383279", "123456789012345678901234567890", "1.5abc", "1d3", "", "-", "--1",
                                   "0.1e-22", "9007199254740993", "4.9e-324", "1.7976931348623157e308", "1e23", "8.5e22"};
static const char* ODD_INDEX[] = {"0", "-1", "+3", "99999999", "18446744073709551615", "18446744073709551616", "123456789012345678901", "1.0", "2e0",
                                  "0x2", "", "a"};
static const char* SEPARATORS[] = {" ", "  ", "\t", " \t ", "\r\n", "\n", "\n\n", " \n", "\v", "\f"};

static void entryLine(text* t, int pat, int integer, ulong r, ulong c) {
    char line[256];
    if (pat) snprintf(line, sizeof line, "%lu %lu\n", r, c);
    else if (integer) snprintf(line, sizeof line, "%lu %lu %ld\n", r, c, (long)below(2000) - 1000);
    else switch (below(4)) {
        case 0:  snprintf(line, sizeof line, "%lu %lu %.17g\n", r, c, (double)(int64_t)rnd() / 9.2e18); break;
        case 1:  snprintf(line, sizeof line, "%lu %lu %.6e\n", r, c, (double)(int64_t)rnd() / 3e12); break;
        case 2:  snprintf(line, sizeof line, "%lu %lu %g\n", r, c, (double)below(100000) / 64.0); break;
        default: snprintf(line, sizeof line, "%lu %lu %.15g\n", r, c, (double)below(1u << 20) * 1e-9); break;
    }
    puts_(t, line);
}

/* a well-formed file; *entriesAt = offset of the entry section */
static void wellFormed(text* t, size_t* entriesAt) {
    const int pat = below(4) == 0, sym = below(3) == 0, integer = !pat && below(4) == 0;
    const ulong M = 1 + below(below(8) == 0 ? 3000 : 40), N = sym ? M : 1 + below(below(8) == 0 ? 3000 : 40);
    const ulong want = below(16) == 0 ? 9000 + below(9000) : below(60);    /* now and then large enough for several chunks */
    text body = {0};
    ulong nz = 0;
    if (below(4)) {
        /* column-major with ascending rows, as the collection's files are (what the conversion's sortedness check expects) */
        const ulong step = M * N / (want + 1) + 1;                         /* mean gap between cells taken */
        for (ulong c = 1; c <= N && nz < want; ++c)
            for (ulong r = (sym ? c : 1) + below(step); r <= M && nz < want; r += 1 + below(2 * step)) { entryLine(&body, pat, integer, r, c); nz++; }
    } else {
        for (; nz < want; ++nz) {                                           /* any order, repeats */
            ulong r = 1 + below(M), c = 1 + below(N);
            if (sym && c > r) { ulong s = r; r = c; c = s; }
            entryLine(&body, pat, integer, r, c);
        }
    }
    char line[256];
    snprintf(line, sizeof line, "%%%%MatrixMarket matrix coordinate %s %s\n", pat ? "pattern" : integer ? "integer" : "real", sym ? "symmetric" : "general");
    puts_(t, line);
    if (below(2)) puts_(t, "% a comment\n%\n");
    if (below(4) == 0) puts_(t, "\n");
    snprintf(line, sizeof line, "%lu %lu %lu\n", M, N, nz);
    puts_(t, line);
    *entriesAt = t->n;
    if (body.n) put(t, body.s, body.n);
    free(body.s);
}

static size_t lineStart(const text* t, size_t from, size_t pos) {
    while (pos > from && t->s[pos - 1] != '\n') --pos;
    return pos;
}
static size_t lineEnd(const text* t, size_t pos) {
    while (pos < t->n && t->s[pos] != '\n') ++pos;
    return pos < t->n ? pos + 1 : t->n;
}
static void splice(text* t, size_t at, size_t del, const char* ins) {
    text out = {0};
    put(&out, t->s, at);
    puts_(&out, ins);
    put(&out, t->s + at + del, t->n - at - del);
    free(t->s);
    *t = out;
}

static void damage(text* t, size_t entriesAt) {
    if (t->n <= entriesAt) entriesAt = 0;
    const size_t span = t->n - entriesAt;
    const size_t pos = entriesAt + below(span ? span : 1);
    switch (below(15)) {
        case 0: t->n = pos; t->s[t->n] = 0; break;                                       /* truncated */
        case 1: if (pos < t->n) t->s[pos] = (char)(below(4) ? 32 + below(95) : below(256)); break;     /* one byte */
        case 2: { size_t a = lineStart(t, entriesAt, pos); splice(t, a, lineEnd(t, pos) - a, ""); break; }   /* a line lost */
        case 3: {                                                                        /* a line twice */
            size_t a = lineStart(t, entriesAt, pos), b = lineEnd(t, pos);
            char* dup = strndup(t->s + a, b - a);
            splice(t, a, 0, dup);
            free(dup);
            break;
        }
        case 4: for (size_t k = entriesAt; k < t->n; ++k) if (t->s[k] == '\n' && below(3) == 0) t->s[k] = ' '; break;   /* lines joined */
        case 5: for (size_t k = entriesAt; k < t->n; ++k) if (t->s[k] == ' ' && below(3) == 0) t->s[k] = '\n'; break;   /* entries split */
        case 6: {                                                                        /* odd separators */
            for (int rep = 0; rep < 8; ++rep) {
                size_t k = entriesAt + below(t->n - entriesAt ? t->n - entriesAt : 1);
                if (k < t->n && (t->s[k] == ' ' || t->s[k] == '\n')) splice(t, k, 1, SEPARATORS[below(sizeof SEPARATORS / sizeof *SEPARATORS)]);
            }
            break;
        }
        case 7: case 8: {                                                                /* a token replaced */
            size_t a = pos;
            while (a > entriesAt && t->s[a - 1] != ' ' && t->s[a - 1] != '\n') --a;
            size_t b = a;
            while (b < t->n && t->s[b] != ' ' && t->s[b] != '\n') ++b;
            const char* with = below(2) ? ODD_VALUES[below(sizeof ODD_VALUES / sizeof *ODD_VALUES)] : ODD_INDEX[below(sizeof ODD_INDEX / sizeof *ODD_INDEX)];
            splice(t, a, b - a, with);
            break;
        }
        case 9: splice(t, lineStart(t, entriesAt, pos), 0, below(2) ? "% in the middle\n" : "\n   \n"); break;
        case 10: puts_(t, below(2) ? "trailing words\n" : "1 1"); break;
        case 11: { size_t a = below(t->n); if (a < t->n) t->s[a] = (char)(32 + below(95)); break; }     /* anywhere, header included */
        case 12: if (t->n && t->s[t->n - 1] == '\n') { t->n--; t->s[t->n] = 0; } break;  /* no newline at the end */
        case 13: {                                                                       /* a size line that lies */
            static const char* COUNTS[] = {"-1", "-29", "768614336404564651", "384307168202282326", "18446744073709551615", "0", "99999999999"};
            if (!entriesAt) break;
            size_t b = entriesAt - 1, a = lineStart(t, 0, b);                            /* "M N NZ" without its newline */
            ulong M = 0, N = 0, nz = 0;
            if (sscanf(t->s + a, "%lu %lu %lu", &M, &N, &nz) != 3) break;
            char line[128];
            const char* lie = COUNTS[below(sizeof COUNTS / sizeof *COUNTS)];
            switch (below(4)) {
                case 0:  snprintf(line, sizeof line, "%s %lu %lu", below(2) ? "-1" : "18446744073709551615", N, nz); break;
                case 1:  snprintf(line, sizeof line, "%lu %s %lu", M, below(2) ? "-1" : "18446744073709551615", nz); break;
                default: snprintf(line, sizeof line, "%lu %lu %s", M, N, lie); break;
            }
            splice(t, a, b - a, line);
            break;
        }
        default: break;                                                                  /* left as it is */
    }
}

/* the serial steps alone, as MMRead takes them when the file is not a regular file */
static MatrixMarket* serialRead(const char* path) {
    FILE* fp = fopen(path, "r");
    if (!fp) return NULL;
    MatrixMarket* out = calloc(1, sizeof *out);
    if (!out) { fclose(fp); return NULL; }
    if (mm_read_banner(fp, &out->mcode) != 0 || MMCheck(out->mcode) || mm_read_mtx_crd_size(fp, &out->M, &out->N, &out->NZ)) goto fail;
    if (!(out->rowLens = calloc(out->M ? out->M : 1, sizeof *out->rowLens))) goto fail;
    {
        const long pos = ftell(fp);
        const int pat = mm_is_pattern(out->mcode), sym = mm_is_symmetric(out->mcode);
        ulong r, c; double v;
        for (;;) {
            int got = pat ? fscanf(fp, "%lu %lu", &r, &c) : fscanf(fp, "%lu %lu %lf", &r, &c, &v);
            if (got == EOF || got != (pat ? 2 : 3)) break;
            if (r > out->M || c > out->N || (sym && (c > out->M || r > out->N))) goto fail;
        }
        if (fseek(fp, pos, SEEK_SET)) goto fail;
    }
    if (!(out->entries = MMtoCOO(&out->NZ, fp, out->mcode, out->rowLens))) goto fail;
    fclose(fp);
    return out;
fail:
    freeMatrixMarket(out);
    fclose(fp);
    return NULL;
}

/* one damaged copy of an archive through the decompressors; returns 1 when something was inflated */
static int damagedArchive(const char* dir, const char* archive) {
    FILE* f = fopen(archive, "rb");
    if (!f) { perror(archive); exit(2); }
    text t = {0};
    char buf[4096];
    for (size_t got; (got = fread(buf, 1, sizeof buf, f)) > 0;) put(&t, buf, got);
    fclose(f);
    for (uint64_t d = 1 + below(3); d > 0 && t.n; --d) switch (below(4)) {
        case 0: t.n = below(t.n); break;                                                 /* cut short */
        case 1: t.s[below(t.n)] = (char)below(256); break;                               /* one byte */
        case 2: { size_t a = below(t.n); for (size_t k = a; k < a + 4 && k < t.n; ++k) t.s[k] = (char)0xFF; break; }
        default: {                                                                       /* near the end: the zip directory, the gzip / xz trailers */
            size_t a = t.n > 64 ? t.n - 1 - below(64) : below(t.n);
            t.s[a] = (char)below(256);
            break;
        }
    }
    const char* ext = strrchr(archive, '.');
    char src[4096], dst[4096];
    snprintf(src, sizeof src, "%s/fuzz.mtx%s", dir, ext ? ext : "");
    snprintf(dst, sizeof dst, "%s/fuzz_inflated.mtx", dir);
    f = fopen(src, "wb");
    if (!f || (t.n && fwrite(t.s, 1, t.n, f) != t.n) || fclose(f)) { perror(src); exit(2); }
    free(t.s);
    const int inflated = extractInTmpFS(src, dst) == 0;
    if (inflated) {
        spmat* a = MMtoCSR(dst);
        if (a) freeSpmat(a);
    }
    remove(src);
    remove(dst);
    return inflated;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <dir> <iterations> <seed> [archives]\n", argv[0]); return 2; }
    const long iterations = atol(argv[2]);
    rngState = strtoull(argv[3], NULL, 0);
    char path[4096];
    snprintf(path, sizeof path, "%s/fuzz.mtx", argv[1]);
    long accepted = 0, rejected = 0, csr = 0, ell = 0, archives = 0, inflated = 0;
    for (long it = 0; it < iterations; ++it) {
        if (argc > 4 && it % 4 == 3) {
            archives++;
            inflated += damagedArchive(argv[1], argv[4 + below((uint64_t)argc - 4)]);
            continue;
        }
        text t = {0};
        size_t entriesAt = 0;
        wellFormed(&t, &entriesAt);
        for (uint64_t d = below(3) + (it % 5 != 0); d > 0; --d) damage(&t, entriesAt);   /* one file in five may stay intact */
        FILE* f = fopen(path, "wb");
        if (!f || fwrite(t.s, 1, t.n, f) != t.n || fclose(f)) { perror(path); return 2; }
        MatrixMarket* fast = MMRead(path);
        MatrixMarket* slow = serialRead(path);
        if (!fast != !slow) {
            printf("DISAGREE (iteration %ld): in-memory parser %s, serial steps %s\n%s\n", it, fast ? "accepted" : "rejected", slow ? "accepted" : "rejected", t.s);
            return 1;
        }
        if (fast) {
            if (fast->M != slow->M || fast->N != slow->N || fast->NZ != slow->NZ ||
                memcmp(fast->entries, slow->entries, fast->NZ * sizeof *fast->entries) ||
                memcmp(fast->rowLens, slow->rowLens, (fast->M ? fast->M : 1) * sizeof *fast->rowLens)) {
                printf("DIFFERENT ENTRIES (iteration %ld)\n%s\n", it, t.s);
                return 1;
            }
            accepted++;
            spmat* a = MMtoCSR(path);
            if (a) { csr++; freeSpmat(a); }
            spmat* e = MMtoELL(path);
            if (e) { ell++; freeSpmat(e); }
        } else rejected++;
        freeMatrixMarket(fast);
        freeMatrixMarket(slow);
        free(t.s);
    }
    remove(path);
    printf("fuzz_loader: %ld files, %ld accepted by both readers (same entries), %ld rejected by both; CSR built %ld, ELL built %ld; "
           "%ld damaged archives, %ld still inflated\n", iterations - archives, accepted, rejected, csr, ell, archives, inflated);
    return 0;
}
