// Host half of the JPEG path: marker parsing + Huffman entropy decoding into quantised DCT coefficients - baseline /
// sequential (ITU-T T.81 Annex F, libjpeg's jdhuff.c) and progressive (Annex G, jdphuff.c: spectral selection and
// successive approximation over several scans, EOB runs, correction bits) - which is what tf.image.decode_jpeg
// (dataset/dataset.py:28) runs first.  Dequantisation, IDCT, upsampling and colour conversion happen on the GPU
// (jpeg_pipeline.hip); a progressive file ends in the same coefficient arrays as a sequential one.
//
// Scope: 8-bit Huffman frames SOF0 / SOF1 / SOF2, 1 or 3 components, sampling factors h,v in {1,2} with the luma plane
// at the maximum, restart intervals, any number of scans (arithmetic SOF9+, lossless, 12-bit, CMYK -> VIP_ERR_JPEG;
// the reference has no fallback either: TF raises).
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "vipcup_hip.h"

void vip_set_error(const char* fmt, ...);

namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t bits[17];
    uint8_t vals[256];
    // canonical decode tables (T.81 F.2.2.3)
    int32_t mincode[17], maxcode[18], valptr[17];
    // 9-bit lookahead: (length << 8) | symbol, 0 = not resolvable in 9 bits
    uint16_t look[512];

    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            k += bits[l];
            code += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        memset(look, 0, sizeof(look));
        code = 0;
        k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int lo = code << (9 - l);
                for (int f = 0; f < (1 << (9 - l)); ++f) look[lo + f] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
};

struct Scan {
    int ns = 0, comp[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0, restart_interval = 0;
    const uint8_t* data = nullptr;
    size_t len = 0;
    HuffTable dc[4], ac[4];   // the tables in force when the scan starts (DHT may be re-sent between scans)
};

struct Parsed {
    bool progressive = false;
    bool saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    bool rgb_coded = false;   // 3 components stored as R,G,B (jdapimin.c default_decompress_parms)
    std::vector<Scan> scans;  // filled for progressive and multi-scan sequential files
    int width = 0, height = 0, ncomp = 0;
    Component comp[3];
    uint16_t qt[4][64];  // natural order
    bool qt_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    const uint8_t* scan = nullptr;  // entropy-coded data
    size_t scan_len = 0;
    int hmax = 1, vmax = 1;
    int mcus_x = 0, mcus_y = 0;
};

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int parse(const uint8_t* d, size_t n, Parsed& P, bool need_scan) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) {
        vip_set_error("jpeg: missing SOI");
        return VIP_ERR_JPEG;
    }
    size_t pos = 2;
    bool have_sof = false;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) {
            vip_set_error("jpeg: marker expected at byte %zu", pos);
            return VIP_ERR_JPEG;
        }
        while (pos < n && d[pos] == 0xFF) ++pos;  // fill bytes
        if (pos >= n) break;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > n) break;
        const int len = rd16(d + pos);
        if (len < 2 || pos + len > n) {
            vip_set_error("jpeg: truncated segment 0x%02X", m);
            return VIP_ERR_JPEG;
        }
        const uint8_t* s = d + pos + 2;
        const int sl = len - 2;
        if (m == 0xE0 && sl >= 5 && memcmp(s, "JFIF\0", 5) == 0) P.saw_jfif = true;
        if (m == 0xEE && sl >= 12 && memcmp(s, "Adobe", 5) == 0) {
            P.saw_adobe = true;
            P.adobe_transform = s[11];
        }
        if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            P.progressive = (m == 0xC2);
            if (sl < 6 || s[0] != 8) {
                vip_set_error("jpeg: only 8-bit precision is supported");
                return VIP_ERR_JPEG;
            }
            P.height = rd16(s + 1);
            P.width = rd16(s + 3);
            P.ncomp = s[5];
            if ((P.ncomp != 1 && P.ncomp != 3) || sl < 6 + 3 * P.ncomp || P.width <= 0 || P.height <= 0) {
                vip_set_error("jpeg: unsupported frame (ncomp=%d %dx%d)", P.ncomp, P.width, P.height);
                return VIP_ERR_JPEG;
            }
            for (int i = 0; i < P.ncomp; ++i) {
                Component& c = P.comp[i];
                c.id = s[6 + 3 * i];
                c.h = s[7 + 3 * i] >> 4;
                c.v = s[7 + 3 * i] & 15;
                c.tq = s[8 + 3 * i] & 3;
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) {
                    vip_set_error("jpeg: sampling factor %dx%d unsupported", c.h, c.v);
                    return VIP_ERR_JPEG;
                }
                if (c.h > P.hmax) P.hmax = c.h;
                if (c.v > P.vmax) P.vmax = c.v;
            }
            if (P.ncomp == 1) {  // a single-component scan is never interleaved: one block per MCU
                P.comp[0].h = P.comp[0].v = 1;
                P.hmax = P.vmax = 1;
            } else if (P.comp[0].h != P.hmax || P.comp[0].v != P.vmax || P.comp[1].h != 1 || P.comp[1].v != 1 ||
                       P.comp[2].h != 1 || P.comp[2].v != 1) {
                vip_set_error("jpeg: only luma-at-full-resolution subsampling (4:4:4, 4:2:2, 4:4:0, 4:2:0) is supported");
                return VIP_ERR_JPEG;
            }
            P.mcus_x = (P.width + 8 * P.hmax - 1) / (8 * P.hmax);
            P.mcus_y = (P.height + 8 * P.vmax - 1) / (8 * P.vmax);
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            vip_set_error("jpeg: SOF%d (arithmetic / lossless / hierarchical) is not supported", m - 0xC0);
            return VIP_ERR_JPEG;
        } else if (m == 0xDB) {
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                ++o;
                if (tq > 3 || o + (pq ? 128 : 64) > sl) {
                    vip_set_error("jpeg: bad DQT");
                    return VIP_ERR_JPEG;
                }
                for (int i = 0; i < 64; ++i) {
                    const int v = pq ? rd16(s + o + 2 * i) : s[o + i];
                    P.qt[tq][ZIGZAG[i]] = (uint16_t)v;
                }
                P.qt_present[tq] = true;
                o += pq ? 128 : 64;
            }
        } else if (m == 0xC4) {
            int o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) {
                    vip_set_error("jpeg: bad DHT");
                    return VIP_ERR_JPEG;
                }
                HuffTable& t = tc ? P.ac[th] : P.dc[th];
                int cnt = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) {
                    t.bits[i] = s[o + i];
                    cnt += t.bits[i];
                }
                o += 17;
                if (cnt > 256 || o + cnt > sl) {
                    vip_set_error("jpeg: bad DHT counts");
                    return VIP_ERR_JPEG;
                }
                memcpy(t.vals, s + o, cnt);
                o += cnt;
                t.present = true;
                t.build();
            }
        } else if (m == 0xDD) {
            if (sl >= 2) P.restart_interval = rd16(s);
        } else if (m == 0xDA) {
            if (!have_sof) {
                vip_set_error("jpeg: SOS before SOF");
                return VIP_ERR_JPEG;
            }
            const int ns = s[0];
            if (ns < 1 || ns > P.ncomp || sl < 1 + 2 * ns + 3) {
                vip_set_error("jpeg: bad SOS header");
                return VIP_ERR_JPEG;
            }
            Scan sc;
            sc.ns = ns;
            for (int i = 0; i < ns; ++i) {
                const int cid = s[1 + 2 * i];
                bool found = false;
                for (int c = 0; c < P.ncomp; ++c)
                    if (P.comp[c].id == cid) {
                        if ((s[2 + 2 * i] >> 4) > 3 || (s[2 + 2 * i] & 15) > 3) {
                            vip_set_error("jpeg: SOS selects Huffman table %d/%d (0..3)", s[2 + 2 * i] >> 4, s[2 + 2 * i] & 15);
                            return VIP_ERR_JPEG;
                        }
                        P.comp[c].td = s[2 + 2 * i] >> 4;
                        P.comp[c].ta = s[2 + 2 * i] & 15;
                        sc.comp[i] = c;
                        sc.td[i] = P.comp[c].td & 3;
                        sc.ta[i] = P.comp[c].ta & 3;
                        found = true;
                    }
                if (!found) {
                    vip_set_error("jpeg: SOS references unknown component");
                    return VIP_ERR_JPEG;
                }
            }
            P.scan = d + pos + len;
            P.scan_len = n - (pos + len);
            if (!P.progressive && ns == P.ncomp) return VIP_OK;      // the single interleaved scan of a sequential file
            if (!need_scan) return VIP_OK;                           // probe: the frame header is all that is needed
            sc.ss = s[1 + 2 * ns];
            sc.se = s[2 + 2 * ns];
            sc.ah = s[3 + 2 * ns] >> 4;
            sc.al = s[3 + 2 * ns] & 15;
            if (!P.progressive) { sc.ss = 0; sc.se = 63; sc.ah = sc.al = 0; }
            if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || (sc.ss == 0 && sc.se != 0 && P.progressive) ||
                (sc.ss > 0 && ns != 1)) {
                vip_set_error("jpeg: bad progressive scan parameters (Ss=%d Se=%d Ah=%d Al=%d Ns=%d)", sc.ss, sc.se, sc.ah, sc.al, ns);
                return VIP_ERR_JPEG;
            }
            sc.restart_interval = P.restart_interval;
            sc.data = d + pos + len;
            for (int t = 0; t < 4; ++t) {
                sc.dc[t] = P.dc[t];
                sc.ac[t] = P.ac[t];
            }
            // skip the entropy-coded segment: up to the next marker that is neither a stuffed 0xFF00 nor RSTn
            size_t q = pos + len;
            while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] != 0x00 && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) ++q;
            sc.len = q - (pos + len);
            P.scans.push_back(sc);
            pos = q;
            continue;
        }
        pos += len;
    }
    if (!have_sof) {
        vip_set_error("jpeg: no SOF0/SOF1/SOF2 frame header");
        return VIP_ERR_JPEG;
    }
    if (!P.scans.empty()) return VIP_OK;
    if (need_scan) {
        vip_set_error("jpeg: no SOS");
        return VIP_ERR_JPEG;
    }
    return VIP_OK;
}

void fill_desc(const Parsed& P, vip_jpeg_desc* d, size_t* elems) {
    memset(d, 0, sizeof(*d));
    d->width = P.width;
    d->height = P.height;
    d->ncomp = P.ncomp;
    // colour space of a 3-component file, as libjpeg guesses it (jdapimin.c:default_decompress_parms): JFIF -> YCbCr;
    // else Adobe transform 0 -> RGB, 1 -> YCbCr; else component ids 'R','G','B' -> RGB; anything else YCbCr
    if (P.ncomp == 3 && !P.saw_jfif) {
        if (P.saw_adobe) d->rgb_coded = (P.adobe_transform == 0);
        else d->rgb_coded = (P.comp[0].id == 'R' && P.comp[1].id == 'G' && P.comp[2].id == 'B');
    }
    size_t off = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        d->hsamp[c] = P.comp[c].h;
        d->vsamp[c] = P.comp[c].v;
        d->blocks_w[c] = P.mcus_x * P.comp[c].h;
        d->blocks_h[c] = P.mcus_y * P.comp[c].v;
        d->coef_off[c] = (int64_t)off;
        off += (size_t)d->blocks_w[c] * d->blocks_h[c] * 64;
        for (int i = 0; i < 64; ++i) d->qt[c][i] = P.qt[P.comp[c].tq][i];
    }
    *elems = off;
}

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;

    inline void refill() {
        while (nbits <= 56) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else {  // a marker (RSTn / EOI): feed zeros from here on, like libjpeg
                        hit_marker = true;
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            acc |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    }
    inline int peek(int n) { return (int)(acc >> (64 - n)); }
    inline void skip(int n) {
        acc <<= n;
        nbits -= n;
    }
    inline int get(int n) {
        if (n == 0) return 0;
        const int v = peek(n);
        skip(n);
        return v;
    }
};

inline int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

inline int decode_sym(BitReader& br, const HuffTable& t) {
    const int look = t.look[br.peek(9)];
    if (look) {
        br.skip(look >> 8);
        return look & 255;
    }
    int code = br.peek(9), l = 9;
    br.skip(9);
    for (;;) {
        ++l;
        if (l > 16) return -1;
        code = (code << 1) | br.get(1);
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
    }
}

// Progressive (and multi-scan sequential) decoding: every scan refines the same coefficient arrays.
// Follows jdphuff.c (decode_mcu_DC_first / DC_refine / AC_first / AC_refine).
int decode_scans(const Parsed& P, const vip_jpeg_desc& D, int16_t* coef) {
    for (const Scan& sc : P.scans) {
        for (int i = 0; i < sc.ns; ++i) {
            const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = sc.se > 0;
            if ((need_dc && !sc.dc[sc.td[i]].present) || (need_ac && !sc.ac[sc.ta[i]].present)) {
                vip_set_error("jpeg: scan uses a Huffman table that was not defined");
                return VIP_ERR_JPEG;
            }
        }
        BitReader br;
        br.p = sc.data;
        br.end = sc.data + sc.len;
        int pred[3] = {0, 0, 0};
        int eobrun = 0;
        const int ss = sc.ss, se = sc.se, al = sc.al;
        const int p1 = 1 << al, m1 = -(1 << al);
        const bool sequential = !P.progressive;

        auto block = [&](int i, int16_t* blk) -> int {
            const int c = sc.comp[i];
            const HuffTable& tdc = sc.dc[sc.td[i]];
            const HuffTable& tac = sc.ac[sc.ta[i]];
            if (sequential) {
                br.refill();
                const int s = decode_sym(br, tdc);
                if (s < 0 || s > 11) return -1;
                br.refill();
                pred[c] += extend(br.get(s), s);
                blk[0] = (int16_t)pred[c];
                for (int k = 1; k < 64;) {
                    br.refill();
                    const int rs = decode_sym(br, tac);
                    if (rs < 0) return -1;
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                        if (r != 15) break;
                        k += 16;
                        continue;
                    }
                    k += r;
                    if (k > 63) return -1;
                    blk[ZIGZAG[k]] = (int16_t)extend(br.get(sz), sz);
                    ++k;
                }
                return 0;
            }
            if (ss == 0) {                                   // DC scan
                br.refill();
                if (sc.ah == 0) {
                    const int s = decode_sym(br, tdc);
                    if (s < 0 || s > 11) return -1;
                    br.refill();
                    pred[c] += extend(br.get(s), s);
                    blk[0] = (int16_t)(pred[c] * (1 << al));
                } else if (br.get(1)) {
                    blk[0] |= (int16_t)p1;
                }
                return 0;
            }
            if (sc.ah == 0) {                                // AC first pass
                if (eobrun > 0) {
                    --eobrun;
                    return 0;
                }
                for (int k = ss; k <= se; ++k) {
                    br.refill();
                    const int rs = decode_sym(br, tac);
                    if (rs < 0) return -1;
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz) {
                        k += r;
                        if (k > 63) return -1;
                        br.refill();
                        blk[ZIGZAG[k]] = (int16_t)(extend(br.get(sz), sz) * (1 << al));
                    } else if (r == 15) {
                        k += 15;
                    } else {
                        eobrun = 1 << r;
                        if (r) {
                            br.refill();
                            eobrun += br.get(r);
                        }
                        --eobrun;
                        break;
                    }
                }
                return 0;
            }
            // AC refinement
            auto refine = [&](int pos) {
                if (blk[pos] != 0) {
                    br.refill();
                    if (br.get(1) && (blk[pos] & p1) == 0) blk[pos] = (int16_t)(blk[pos] + (blk[pos] >= 0 ? p1 : m1));
                }
            };
            int k = ss;
            if (eobrun == 0) {
                for (; k <= se; ++k) {
                    br.refill();
                    const int rs = decode_sym(br, tac);
                    if (rs < 0) return -1;
                    int r = rs >> 4;
                    const int sz = rs & 15;
                    int val = 0;
                    if (sz) {
                        br.refill();
                        val = br.get(1) ? p1 : m1;
                    } else if (r != 15) {
                        eobrun = 1 << r;
                        if (r) {
                            br.refill();
                            eobrun += br.get(r);
                        }
                        break;
                    }
                    while (k <= se) {
                        const int pos = ZIGZAG[k];
                        if (blk[pos] != 0) {
                            refine(pos);
                        } else if (--r < 0) {
                            break;
                        }
                        ++k;
                    }
                    if (val) {
                        if (k > 63) return -1;
                        blk[ZIGZAG[k]] = (int16_t)val;
                    }
                }
            }
            if (eobrun > 0) {
                for (; k <= se; ++k) refine(ZIGZAG[k]);
                --eobrun;
            }
            return 0;
        };

        int to_restart = sc.restart_interval;
        auto restart_if_due = [&]() {
            if (sc.restart_interval && to_restart == 0) {
                br.acc = 0;
                br.nbits = 0;
                br.hit_marker = false;
                if (br.p + 1 < br.end && br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7) br.p += 2;
                pred[0] = pred[1] = pred[2] = 0;
                eobrun = 0;
                to_restart = sc.restart_interval;
            }
        };
        if (sc.ns > 1) {                                     // interleaved scan: MCU order
            for (int my = 0; my < P.mcus_y; ++my)
                for (int mx = 0; mx < P.mcus_x; ++mx) {
                    restart_if_due();
                    for (int i = 0; i < sc.ns; ++i) {
                        const int c = sc.comp[i];
                        for (int by = 0; by < P.comp[c].v; ++by)
                            for (int bx = 0; bx < P.comp[c].h; ++bx) {
                                const long brow = (long)my * P.comp[c].v + by, bcol = (long)mx * P.comp[c].h + bx;
                                if (block(i, coef + D.coef_off[c] + (brow * D.blocks_w[c] + bcol) * 64) != 0) {
                                    vip_set_error("jpeg: corrupt Huffman code in a progressive scan");
                                    return VIP_ERR_JPEG;
                                }
                            }
                    }
                    if (sc.restart_interval) --to_restart;
                }
        } else {                                             // one component: its own raster of ceil(size / 8) blocks
            const int c = sc.comp[0];
            const int cw = (P.width * P.comp[c].h + P.hmax - 1) / P.hmax, chh = (P.height * P.comp[c].v + P.vmax - 1) / P.vmax;
            const int bw = (cw + 7) / 8, bh = (chh + 7) / 8;
            for (int y = 0; y < bh; ++y)
                for (int x = 0; x < bw; ++x) {
                    restart_if_due();
                    if (block(0, coef + D.coef_off[c] + ((long)y * D.blocks_w[c] + x) * 64) != 0) {
                        vip_set_error("jpeg: corrupt Huffman code in a progressive scan");
                        return VIP_ERR_JPEG;
                    }
                    if (sc.restart_interval) --to_restart;
                }
        }
    }
    return VIP_OK;
}

int decode_image(const Parsed& P, const vip_jpeg_desc& D, int16_t* coef) {
    if (!P.scans.empty()) {
        for (int c = 0; c < P.ncomp; ++c)
            if (!P.qt_present[P.comp[c].tq]) {
                vip_set_error("jpeg: missing quantisation table");
                return VIP_ERR_JPEG;
            }
        return decode_scans(P, D, coef);
    }
    for (int c = 0; c < P.ncomp; ++c) {
        if (!P.qt_present[P.comp[c].tq] || !P.dc[P.comp[c].td].present || !P.ac[P.comp[c].ta].present) {
            vip_set_error("jpeg: missing quantisation / Huffman table");
            return VIP_ERR_JPEG;
        }
    }
    BitReader br;
    br.p = P.scan;
    br.end = P.scan + P.scan_len;
    int pred[3] = {0, 0, 0};
    int to_restart = P.restart_interval;
    for (int my = 0; my < P.mcus_y; ++my) {
        for (int mx = 0; mx < P.mcus_x; ++mx) {
            if (P.restart_interval && to_restart == 0) {
                // byte-align, expect RSTn
                // the reader never passes a marker, so p sits on the RSTn whether or not refill() saw it
                br.acc = 0;
                br.nbits = 0;
                br.hit_marker = false;
                if (br.p + 1 < br.end && br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7) br.p += 2;
                pred[0] = pred[1] = pred[2] = 0;
                to_restart = P.restart_interval;
            }
            for (int c = 0; c < P.ncomp; ++c) {
                const HuffTable& tdc = P.dc[P.comp[c].td];
                const HuffTable& tac = P.ac[P.comp[c].ta];
                for (int by = 0; by < P.comp[c].v; ++by) {
                    for (int bx = 0; bx < P.comp[c].h; ++bx) {
                        const long brow = (long)my * P.comp[c].v + by, bcol = (long)mx * P.comp[c].h + bx;
                        int16_t* blk = coef + D.coef_off[c] + (brow * D.blocks_w[c] + bcol) * 64;
                        br.refill();
                        int s = decode_sym(br, tdc);
                        if (s < 0 || s > 11) {
                            vip_set_error("jpeg: corrupt DC code");
                            return VIP_ERR_JPEG;
                        }
                        br.refill();
                        pred[c] += extend(br.get(s), s);
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            br.refill();
                            const int rs = decode_sym(br, tac);
                            if (rs < 0) {
                                vip_set_error("jpeg: corrupt AC code");
                                return VIP_ERR_JPEG;
                            }
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break;  // EOB
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) {
                                vip_set_error("jpeg: AC run past the end of the block");
                                return VIP_ERR_JPEG;
                            }
                            blk[ZIGZAG[k]] = (int16_t)extend(br.get(sz), sz);
                            ++k;
                        }
                    }
                }
            }
            if (P.restart_interval) --to_restart;
        }
    }
    return VIP_OK;
}

}  // namespace

extern "C" int vip_jpeg_probe_h(const uint8_t* jpeg_h, size_t len, vip_jpeg_desc* desc_h, size_t* coef_elems_h) {
    if (!jpeg_h || !desc_h || !coef_elems_h) {
        vip_set_error("vip_jpeg_probe_h: null pointer");
        return VIP_ERR_BAD_ARG;
    }
    Parsed P;
    const int st = parse(jpeg_h, len, P, false);
    if (st != VIP_OK) return st;
    fill_desc(P, desc_h, coef_elems_h);
    return VIP_OK;
}

extern "C" int vip_jpeg_entropy_decode_h(const uint8_t* const* jpeg_h, const size_t* len_h, int n,
                                         vip_jpeg_desc* desc_h, int16_t* coef_h, size_t coef_cap,
                                         size_t* coef_used_h, int threads) {
    if (!jpeg_h || !len_h || !desc_h || !coef_h || n < 0) {
        vip_set_error("vip_jpeg_entropy_decode_h: bad argument");
        return VIP_ERR_BAD_ARG;
    }
    // pass 1 (serial, cheap): headers -> descriptors and coefficient offsets
    std::vector<Parsed> parsed(n);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const int st = parse(jpeg_h[i], len_h[i], parsed[i], true);
        if (st != VIP_OK) return st;
        size_t elems = 0;
        fill_desc(parsed[i], &desc_h[i], &elems);
        for (int c = 0; c < parsed[i].ncomp; ++c) desc_h[i].coef_off[c] += (int64_t)off;
        off += elems;
    }
    if (coef_used_h) *coef_used_h = off;
    if (off > coef_cap) {
        vip_set_error("vip_jpeg_entropy_decode_h: coefficient buffer too small (%zu > %zu)", off, coef_cap);
        return VIP_ERR_BAD_ARG;
    }
    memset(coef_h, 0, off * sizeof(int16_t));
    // pass 2: entropy decoding, one image at a time per worker
    if (threads < 1) threads = 1;
    if (threads > n) threads = n > 0 ? n : 1;
    std::atomic<int> next(0), status(VIP_OK);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || status.load() != VIP_OK) return;
            const int st = decode_image(parsed[i], desc_h[i], coef_h);
            if (st != VIP_OK) status.store(st);
        }
    };
    if (threads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    return status.load();
}
