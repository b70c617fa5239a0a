/* kseq_dump.c -- drives the REFERENCE's FASTA reader (src/kseq.h, compiled from where it
 * lies) exactly as load_seqs does (src/pairsnp.hpp:60,75-101): KSEQ_INIT(gzFile, gzread),
 * gzopen, kseq_read until < 0.  Prints "name<TAB>sequence" per record and the final return
 * code.  TEST INFRASTRUCTURE ONLY: pins oracle/tracs_oracle.c:orc_read_fasta.            */
#include <stdio.h>
#include <zlib.h>
#include "kseq.h"
KSEQ_INIT(gzFile, gzread)

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    int quiet = argc > 2 && argv[2][0] == '-' && argv[2][1] == 'q';   /* -q: parse only (timing the reference reader) */
    gzFile fp = gzopen(argv[1], "r");
    if (!fp) { printf("#rc=-5\n"); return 0; }
    kseq_t *seq = kseq_init(fp);
    int l;
    long nrec = 0, nbytes = 0;
    while ((l = kseq_read(seq)) >= 0) {
        nrec++; nbytes += l;
        if (!quiet) printf("%s\t%s\n", seq->name.s ? seq->name.s : "", (seq->seq.s && l > 0) ? seq->seq.s : "");
    }
    if (quiet) printf("#records=%ld bases=%ld\n", nrec, nbytes);
    printf("#rc=%d\n", l);
    kseq_destroy(seq);
    gzclose(fp);
    return 0;
}
