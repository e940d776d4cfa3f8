/* bn_flac.c — FLAC stream decoder for the ingest side of the path (host code, plain C, no dependencies).
 *
 * The reference reads every container through libsndfile (`soundfile.read`, reference: birdnet_stm32/audio/io.py:90,114-116;
 * its dataset walker lists .wav/.flac/.ogg/.mp3/.m4a, data/dataset.py).  libsndfile is not on the MI355X image, so besides RIFF/WAVE
 * (parsed in Python, audio/io.py) this build decodes native FLAC itself: the lossless format bird-sound archives ship most.
 * Written from the format specification (RFC 9639): STREAMINFO, frame headers (fixed and variable block size, UTF-8 coded numbers,
 * CRC-8), CONSTANT / VERBATIM / FIXED / LPC subframes, Rice and Rice2 residuals with escape partitions, wasted bits, the three stereo
 * decorrelation modes, CRC-16 per frame.  Output: interleaved int32 samples at the stream's own bit depth (the caller scales by
 * 2^-(bps-1) like libsndfile).  The decoded audio's MD5 (STREAMINFO, bn_flac_md5) is checked by the Python wrapper whenever it decodes a
 * whole stream (audio/_flac.py: decode_flac); a frame whose bit depth differs from STREAMINFO's is refused.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BN_FLAC_ERR_FORMAT (-1)
#define BN_FLAC_ERR_CRC (-2)
#define BN_FLAC_ERR_UNSUPPORTED (-3)
#define BN_FLAC_ERR_NOMEM (-4)

typedef struct {
    const uint8_t* p;
    size_t n, pos;   /* byte position */
    uint64_t acc;    /* bit accumulator, `bits` valid low bits */
    int bits;
    int eof;
} BitReader;

static void br_init(BitReader* b, const uint8_t* p, size_t n, size_t pos) {
    b->p = p; b->n = n; b->pos = pos; b->acc = 0; b->bits = 0; b->eof = 0;
}
static uint32_t br_read(BitReader* b, int k) { /* k <= 32 */
    if (k == 0) return 0;
    while (b->bits < k) {
        uint64_t byte = 0;
        if (b->pos < b->n) byte = b->p[b->pos++];
        else b->eof = 1;
        b->acc = (b->acc << 8) | byte;
        b->bits += 8;
    }
    b->bits -= k;
    return (uint32_t)((b->acc >> b->bits) & ((k == 32) ? 0xffffffffull : ((1ull << k) - 1)));
}
static int32_t br_read_signed(BitReader* b, int k) {
    if (k == 0) return 0;
    uint32_t v = br_read(b, k);
    if (k < 32 && (v >> (k - 1))) v |= ~0u << k;
    return (int32_t)v;
}
static uint32_t br_unary(BitReader* b) { /* zeros before the next one */
    uint32_t q = 0;
    while (!b->eof && br_read(b, 1) == 0) ++q;
    return q;
}
static void br_align(BitReader* b) { b->bits -= b->bits & 7; }
static size_t br_bytepos(const BitReader* b) { return b->pos - (size_t)(b->bits >> 3); }

static uint8_t crc8(const uint8_t* p, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
    }
    return c;
}
static uint16_t crc16(const uint8_t* p, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint16_t)(p[i] << 8);
        for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
    }
    return c;
}

typedef struct {
    int sample_rate, channels, bps, min_block, max_block;
    int64_t total;
    size_t first_frame; /* byte offset of the first audio frame */
    uint8_t md5[16];    /* MD5 of the decoded audio as the encoder wrote it (all zero: none) */
} StreamInfo;

static int parse_header(const uint8_t* d, size_t n, StreamInfo* si) {
    size_t pos = 0;
    if (n >= 10 && memcmp(d, "ID3", 3) == 0) { /* ID3v2 tag in front of the stream */
        size_t sz = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (d[9] & 0x7f);
        pos = 10 + sz;
    }
    if (pos + 4 > n || memcmp(d + pos, "fLaC", 4) != 0) return BN_FLAC_ERR_FORMAT;
    pos += 4;
    int have = 0;
    for (;;) {
        if (pos + 4 > n) return BN_FLAC_ERR_FORMAT;
        const int last = d[pos] >> 7, type = d[pos] & 0x7f;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos + len > n) return BN_FLAC_ERR_FORMAT;
        if (type == 0) {
            if (len < 34) return BN_FLAC_ERR_FORMAT;
            const uint8_t* s = d + pos;
            si->min_block = (s[0] << 8) | s[1];
            si->max_block = (s[2] << 8) | s[3];
            si->sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si->channels = ((s[12] >> 1) & 7) + 1;
            si->bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si->total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            memcpy(si->md5, s + 18, 16);
            have = 1;
        }
        pos += len;
        if (last) break;
    }
    if (!have || si->sample_rate <= 0 || si->bps < 4 || si->bps > 32) return BN_FLAC_ERR_FORMAT;
    si->first_frame = pos;
    return 0;
}

static int decode_residual(BitReader* b, int32_t* r, int blocksize, int order) {
    const int method = (int)br_read(b, 2);
    if (method > 1) return BN_FLAC_ERR_FORMAT;
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int po = (int)br_read(b, 4);
    const int parts = 1 << po;
    if ((blocksize >> po) << po != blocksize && po > 0) return BN_FLAC_ERR_FORMAT;
    if ((blocksize >> po) < order) return BN_FLAC_ERR_FORMAT;
    int idx = 0;
    for (int p = 0; p < parts; ++p) {
        const int cnt = (blocksize >> po) - (p == 0 ? order : 0);
        const int param = (int)br_read(b, pbits);
        if (param == esc) {
            const int nb = (int)br_read(b, 5);
            for (int i = 0; i < cnt; ++i) r[idx++] = br_read_signed(b, nb);
        } else {
            for (int i = 0; i < cnt; ++i) {
                const uint32_t q = br_unary(b);
                const uint32_t u = (q << param) | br_read(b, param);
                r[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (b->eof) return BN_FLAC_ERR_FORMAT;
    }
    return 0;
}

static int decode_subframe(BitReader* b, int32_t* s, int blocksize, int bps) {
    if (br_read(b, 1)) return BN_FLAC_ERR_FORMAT;
    const int type = (int)br_read(b, 6);
    int wasted = 0;
    if (br_read(b, 1)) wasted = (int)br_unary(b) + 1;
    if (wasted > 31) return BN_FLAC_ERR_FORMAT; /* (a shift by 32 of the 32-bit sample would be undefined) */
    bps -= wasted;
    if (bps < 1 || bps > 33) return BN_FLAC_ERR_FORMAT;
    if (bps > 32) return BN_FLAC_ERR_UNSUPPORTED; /* 32-bit side channels */
    if (type == 0) {
        const int32_t v = br_read_signed(b, bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) s[i] = br_read_signed(b, bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return BN_FLAC_ERR_FORMAT;
        for (int i = 0; i < order; ++i) s[i] = br_read_signed(b, bps);
        int rc = decode_residual(b, s + order, blocksize, order);
        if (rc) return rc;
        for (int i = order; i < blocksize; ++i) {
            int64_t pred = 0;
            switch (order) {
                case 1: pred = s[i - 1]; break;
                case 2: pred = 2 * (int64_t)s[i - 1] - s[i - 2]; break;
                case 3: pred = 3 * (int64_t)s[i - 1] - 3 * (int64_t)s[i - 2] + s[i - 3]; break;
                case 4: pred = 4 * (int64_t)s[i - 1] - 6 * (int64_t)s[i - 2] + 4 * (int64_t)s[i - 3] - s[i - 4]; break;
                default: break;
            }
            s[i] = (int32_t)(pred + s[i]);
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > blocksize) return BN_FLAC_ERR_FORMAT;
        for (int i = 0; i < order; ++i) s[i] = br_read_signed(b, bps);
        const int prec = (int)br_read(b, 4) + 1;
        if (prec == 16) return BN_FLAC_ERR_FORMAT;
        const int shift = br_read_signed(b, 5);
        if (shift < 0) return BN_FLAC_ERR_UNSUPPORTED;
        int32_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br_read_signed(b, prec);
        int rc = decode_residual(b, s + order, blocksize, order);
        if (rc) return rc;
        for (int i = order; i < blocksize; ++i) {
            int64_t sum = 0;
            for (int k = 0; k < order; ++k) sum += (int64_t)coef[k] * s[i - 1 - k];
            s[i] = (int32_t)((sum >> shift) + s[i]);
        }
    } else {
        return BN_FLAC_ERR_FORMAT; /* reserved subframe types */
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) s[i] = (int32_t)((uint32_t)s[i] << wasted);
    return b->eof ? BN_FLAC_ERR_FORMAT : 0;
}

int bn_flac_info(const uint8_t* data, size_t n, int* sample_rate, int* channels, int* bps, int64_t* total_frames) {
    StreamInfo si;
    int rc = parse_header(data, n, &si);
    if (rc) return rc;
    *sample_rate = si.sample_rate; *channels = si.channels; *bps = si.bps; *total_frames = si.total;
    return 0;
}

/* The 16 bytes of STREAMINFO's MD5 (all zero = the encoder did not write one) as THIS parser locates them: behind an ID3v2 tag and at
 * the stream marker the header parser accepted, not at the first "fLaC" byte pattern in the file. */
int bn_flac_md5(const uint8_t* data, size_t n, uint8_t* md5_out) {
    StreamInfo si;
    int rc = parse_header(data, n, &si);
    if (rc) return rc;
    memcpy(md5_out, si.md5, 16);
    return 0;
}

/* Decode frames [first, first + max_frames) (inter-channel sample frames) into out[frames][channels]; returns the number of frames
 * written or a negative error.  A stream whose STREAMINFO holds total = 0 (unknown) is decoded to its end.  out == NULL: nothing is stored,
 * the frames are only counted (the length of a stream that does not state it: frames carry no byte length, they have to be decoded). */
int64_t bn_flac_decode(const uint8_t* data, size_t n, int64_t first, int64_t max_frames, int32_t* out) {
    StreamInfo si;
    int rc = parse_header(data, n, &si);
    if (rc) return rc;
    const int ch = si.channels;
    int32_t* buf = (int32_t*)malloc((size_t)ch * 65536 * sizeof(int32_t));
    if (!buf) return BN_FLAC_ERR_NOMEM;
    size_t pos = si.first_frame;
    int64_t at = 0, written = 0;
    static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int kBps[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    while (pos + 6 <= n && written < max_frames) {
        if (data[pos] != 0xff || (data[pos + 1] & 0xfe) != 0xf8) { rc = BN_FLAC_ERR_FORMAT; break; }
        BitReader b;
        br_init(&b, data, n, pos + 2);
        const int bs_code = (int)br_read(&b, 4), sr_code = (int)br_read(&b, 4), ca = (int)br_read(&b, 4), ss_code = (int)br_read(&b, 3);
        if (br_read(&b, 1)) { rc = BN_FLAC_ERR_FORMAT; break; }
        /* UTF-8 style coded frame / sample number: skip its continuation bytes */
        {
            uint32_t lead = br_read(&b, 8);
            int extra = 0;
            if (lead & 0x80) {
                while (lead & (0x40u >> extra)) ++extra;
                ++extra;
                if (extra > 7) { rc = BN_FLAC_ERR_FORMAT; break; }
                extra -= 1;
            }
            for (int i = 0; i < extra; ++i) (void)br_read(&b, 8);
        }
        int blocksize = kBlock[bs_code];
        if (bs_code == 6) blocksize = (int)br_read(&b, 8) + 1;
        else if (bs_code == 7) blocksize = (int)br_read(&b, 16) + 1;
        if (sr_code == 12) (void)br_read(&b, 8);
        else if (sr_code == 13 || sr_code == 14) (void)br_read(&b, 16);
        else if (sr_code == 15) { rc = BN_FLAC_ERR_FORMAT; break; }
        const size_t hdr_end = br_bytepos(&b);
        const uint32_t want8 = br_read(&b, 8);
        if (blocksize <= 0 || blocksize > 65535 || b.eof) { rc = BN_FLAC_ERR_FORMAT; break; }
        if (crc8(data + pos, hdr_end - pos) != want8) { rc = BN_FLAC_ERR_CRC; break; }
        int bps = ss_code ? kBps[ss_code] : si.bps;
        /* the wrapper scales every sample by STREAMINFO's 2^-(bps-1): a frame that states another depth would be mis-scaled silently */
        if (bps == 0 || bps != si.bps) { rc = BN_FLAC_ERR_FORMAT; break; }
        const int nch = ca < 8 ? ca + 1 : 2;
        if (nch != ch || ca > 10) { rc = BN_FLAC_ERR_FORMAT; break; }
        for (int c = 0; c < nch && !rc; ++c) {
            const int side = (ca == 8 && c == 1) || (ca == 9 && c == 0) || (ca == 10 && c == 1);
            rc = decode_subframe(&b, buf + (size_t)c * 65536, blocksize, bps + side);
        }
        if (rc) break;
        br_align(&b);
        const size_t body_end = br_bytepos(&b);
        const uint32_t want16 = br_read(&b, 16);
        if (b.eof) { rc = BN_FLAC_ERR_FORMAT; break; }
        if (crc16(data + pos, body_end - pos) != want16) { rc = BN_FLAC_ERR_CRC; break; }
        int32_t* c0 = buf;
        int32_t* c1 = buf + 65536;
        if (ca == 8) for (int i = 0; i < blocksize; ++i) c1[i] = c0[i] - c1[i];
        else if (ca == 9) for (int i = 0; i < blocksize; ++i) c0[i] = c0[i] + c1[i];
        else if (ca == 10)
            for (int i = 0; i < blocksize; ++i) {
                const int32_t side = c1[i];
                const int64_t mid = ((int64_t)c0[i] << 1) | (side & 1);
                c0[i] = (int32_t)((mid + side) >> 1);
                c1[i] = (int32_t)((mid - side) >> 1);
            }
        for (int i = 0; i < blocksize && written < max_frames; ++i, ++at) {
            if (at < first) continue;
            if (out)
                for (int c = 0; c < ch; ++c) out[written * ch + c] = buf[(size_t)c * 65536 + i];
            ++written;
        }
        pos = body_end + 2;
    }
    free(buf);
    if (rc) return rc;
    return written;
}
