// fuzz_parsers.cpp -- CPU-only robustness harness for the container / header parsers of the product
// (ultragroth_amd/csrc/host_util.cpp: BinFile, loadZkeyHeader, loadWtnsHeader), built with
// -fsanitize=address,undefined by tests/test_parsers_sanitized.py. The .wtns buffer of a proving service is untrusted
// input: whatever bytes arrive, parsing must end in a normal return or a C++ exception -- never in an out-of-bounds read.
//
// usage: fuzz_parsers <file> <zkey|wtns> <iterations> <seed>
// Mutations of the given file: bit flips in the first 4 KiB and around every section header, truncations, section sizes
// replaced by extreme values (0, 2^32, 2^63, 2^64 - k). After a successful parse every section the prover would touch is
// read through to its claimed end (one byte per 4 KiB and the last byte), which is what trips the sanitizer if a size lies.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <string>
#include <vector>
#include "host_util.hpp"

using namespace ughost;

static uint64_t rng_state;
static uint64_t rnd() {                       // splitmix64
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static volatile uint8_t sink;
static void touch(const BinFile& f, uint32_t id) {
    if (!f.hasSection(id)) return;
    const uint8_t* p = f.sectionData(id);
    uint64_t n = f.sectionSize(id);
    for (uint64_t i = 0; i < n; i += 4096) sink = p[i];
    if (n) sink = p[n - 1];
}

static int parse(const std::vector<uint8_t>& buf, bool zkey) {
    try {
        BinFile f(buf.data(), buf.size(), zkey ? "zkey" : "wtns", zkey ? 1 : 2);
        if (zkey) {
            ZkeyHeader h = loadZkeyHeader(f, false);
            if (h.rIsBn254) { sink = h.alpha1[0]; sink = h.delta2[127]; }
            for (uint32_t id = 1; id <= 12; id++) touch(f, id);
        } else {
            WtnsHeader h = loadWtnsHeader(f);
            (void)h;
            for (uint32_t id = 1; id <= 6; id++) touch(f, id);
        }
        return 0;
    } catch (const std::exception&) {
        return 1;
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: fuzz_parsers <file> <zkey|wtns> <iterations> <seed>\n"); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    std::vector<uint8_t> orig((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const bool zkey = std::string(argv[2]) == "zkey";
    const long iters = atol(argv[3]);
    rng_state = strtoull(argv[4], nullptr, 10);
    if (parse(orig, zkey) != 0) { fprintf(stderr, "the unmodified file does not parse\n"); return 3; }
    // offsets of the section headers of the original file (type u32, size u64 at +4)
    std::vector<size_t> hdrs;
    {
        uint32_t n; memcpy(&n, orig.data() + 8, 4);
        size_t pos = 12;
        for (uint32_t i = 0; i < n && pos + 12 <= orig.size(); i++) {
            hdrs.push_back(pos);
            uint64_t sz; memcpy(&sz, orig.data() + pos + 4, 8);
            pos += 12 + sz;
        }
    }
    const uint64_t extremes[] = {0, 1, 0xffffffffull, 0x100000000ull, 0x7fffffffffffffffull, 0x8000000000000000ull,
                                 0xffffffffffffffffull, 0xfffffffffffffff4ull, 0xffffffffffffffe8ull};
    long ok = 0, rejected = 0;
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> buf = orig;
        switch (rnd() % 5) {
            case 0: {                                  // bit flips near the start (magic, version, counts, first headers)
                int k = 1 + (int)(rnd() % 4);
                for (int j = 0; j < k; j++) { size_t o = rnd() % (buf.size() < 4096 ? buf.size() : 4096); buf[o] ^= (uint8_t)(1u << (rnd() % 8)); }
                break;
            }
            case 1: {                                  // bit flips inside a section header
                size_t h = hdrs[rnd() % hdrs.size()];
                buf[h + rnd() % 12] ^= (uint8_t)(1u << (rnd() % 8));
                break;
            }
            case 2: buf.resize(rnd() % (buf.size() + 1)); break;           // truncation
            case 3: {                                  // a section size replaced by an extreme value (+- a small offset)
                size_t h = hdrs[rnd() % hdrs.size()];
                uint64_t v = extremes[rnd() % (sizeof extremes / sizeof extremes[0])] + (rnd() % 3) - 1;
                memcpy(buf.data() + h + 4, &v, 8);
                break;
            }
            default: {                                 // header fields of the first payload section (n8, counts)
                if (hdrs.size() > 1) {
                    size_t h = hdrs[1] + 12;
                    if (h + 64 < buf.size()) { uint32_t v = (uint32_t)rnd(); memcpy(buf.data() + h + (rnd() % 16) * 4, &v, 4); }
                }
                break;
            }
        }
        if (parse(buf, zkey) == 0) ok++; else rejected++;
    }
    printf("%ld parsed, %ld rejected, 0 crashed\n", ok, rejected);
    return 0;
}
