// host_util.cpp -- see host_util.hpp.
#include "host_util.hpp"
#include <cstdlib>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <system_error>
#include <unistd.h>

namespace ughost {

namespace {
uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

const uint8_t BN254_R_LE[32] = {
    0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
    0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
}  // namespace

// ---- BinFile: magic[4] | u32 version | u32 nSections | { u32 type, u64 size, bytes }* ----------------
BinFile::BinFile(const void* data, uint64_t size, const std::string& type, uint32_t maxVersion) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    if (size < 12) throw std::range_error("File is too short.");
    std::string got(reinterpret_cast<const char*>(p), 4);
    if (got != type) throw std::invalid_argument("Invalid file type. It should be " + type + " and it is " + got);
    uint32_t version = rd32(p + 4);
    if (version > maxVersion)
        throw std::invalid_argument("Invalid version. It should be <=" + std::to_string(maxVersion) + " and it is " + std::to_string(version));
    uint32_t nSections = rd32(p + 8);
    if (size < 12 + (uint64_t)nSections * 12)
        throw std::range_error("File is too short to contain " + std::to_string(nSections) + " sections.");
    uint64_t pos = 12;
    for (uint32_t i = 0; i < nSections; i++) {
        if (pos + 12 > size)
            throw std::range_error("File pos is too big. There are " + std::to_string(size) + " bytes and it's trying to access byte " + std::to_string(pos + 12));
        uint32_t sType = rd32(p + pos);
        uint64_t sSize = rd64(p + pos + 4);
        pos += 12;
        // checked before the add: a crafted size near 2^64 must not wrap `pos` back into the buffer (the reference adds
        // first, binfile_utils.cpp:60-66, and would accept it)
        if (sSize > size - pos)
            throw std::range_error("Section #" + std::to_string(i) + " is invalid.. It ends at pos " +
                                   (sSize > UINT64_MAX - pos ? "beyond 2^64" : std::to_string(pos + sSize)) +
                                   " but should end before " + std::to_string(size) + ".");
        sections_[sType].push_back(Section{p + pos, sSize});
        pos += sSize;
    }
}
const Section& BinFile::find(uint32_t id, uint32_t pos) const {
    auto it = sections_.find(id);
    if (it == sections_.end()) throw std::range_error("Section does not exist: " + std::to_string(id));
    if (pos >= it->second.size())
        throw std::range_error("Section pos too big. There are " + std::to_string(it->second.size()) +
                               " and it's trying to access section: " + std::to_string(pos));
    return it->second[pos];
}
const uint8_t* BinFile::sectionData(uint32_t id, uint32_t pos) const { return find(id, pos).start; }
uint64_t BinFile::sectionSize(uint32_t id, uint32_t pos) const { return find(id, pos).size; }

// ---- FileMap ---------------------------------------------------------------------------------------------
FileMap::FileMap(const std::string& path) {
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ == -1) throw std::system_error(errno, std::generic_category(), "open");
    struct stat sb;
    if (fstat(fd_, &sb) == -1) { int e = errno; close(fd_); fd_ = -1; throw std::system_error(e, std::generic_category(), "fstat"); }
    size_ = (uint64_t)sb.st_size;
    addr_ = mmap(nullptr, size_ ? size_ : 1, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (addr_ == MAP_FAILED) { int e = errno; close(fd_); fd_ = -1; addr_ = nullptr; throw std::system_error(e, std::generic_category(), "mmap failed"); }
    madvise(addr_, size_, MADV_SEQUENTIAL);
}
FileMap::~FileMap() {
    if (addr_) munmap(addr_, size_ ? size_ : 1);
    if (fd_ != -1) close(fd_);
}

// ---- headers ---------------------------------------------------------------------------------------------
namespace {
struct Reader {           // bounds-checked cursor over one section
    const uint8_t* p; uint64_t left;
    const uint8_t* take(uint64_t n) {
        if (n > left) throw std::range_error("Invalid section size");
        const uint8_t* r = p; p += n; left -= n; return r;
    }
    uint32_t u32() { return rd32(take(4)); }
};
}  // namespace

ZkeyHeader loadZkeyHeader(const BinFile& f, bool ultra) {
    ZkeyHeader h;
    Reader s1{f.sectionData(1), f.sectionSize(1)};
    uint32_t protocol = s1.u32();
    if (!ultra && protocol != 1) throw std::invalid_argument("zkey file is not groth16");
    if (ultra && protocol != 1337) throw std::invalid_argument("zkey file is not ultragroth");
    Reader s{f.sectionData(2), f.sectionSize(2)};
    h.n8q = s.u32();
    s.take(h.n8q);
    h.n8r = s.u32();
    const uint8_t* rPrime = s.take(h.n8r);
    h.rIsBn254 = (h.n8r == 32 && h.n8q == 32 && memcmp(rPrime, BN254_R_LE, 32) == 0);
    h.nVars = s.u32();
    h.nPublic = s.u32();
    h.domainSize = s.u32();
    if (ultra) { h.numIndexesC1 = s.u32(); h.numIndexesC2 = s.u32(); h.randIndx = s.u32(); }
    if (!h.rIsBn254) return h;                     // caller reports "zkey curve not supported"
    h.alpha1 = s.take(64);
    h.beta1 = s.take(64);
    h.beta2 = s.take(128);
    h.gamma2 = s.take(128);
    if (ultra) { h.roundDelta1 = s.take(64); h.roundDelta2 = s.take(128); }
    h.delta1 = s.take(64);
    h.delta2 = s.take(128);
    h.nCoefs = f.sectionSize(4) / (12 + h.n8r);
    return h;
}

WtnsHeader loadWtnsHeader(const BinFile& f) {
    WtnsHeader h;
    Reader s{f.sectionData(1), f.sectionSize(1)};
    h.n8 = s.u32();
    const uint8_t* prime = s.take(h.n8);
    h.primeIsBn254 = (h.n8 == 32 && memcmp(prime, BN254_R_LE, 32) == 0);
    h.nVars = s.u32();
    return h;
}

// ---- decimal ---------------------------------------------------------------------------------------------
std::string toDecimal(const uint8_t le[32]) {
    uint32_t w[8];
    memcpy(w, le, 32);
    std::string out;
    bool nonzero = false;
    for (int i = 0; i < 8; i++) nonzero |= (w[i] != 0);
    if (!nonzero) return "0";
    while (true) {
        nonzero = false;
        uint64_t rem = 0;
        for (int i = 7; i >= 0; i--) {               // divide by 10^9
            uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 1000000000u);
            rem = cur % 1000000000u;
            nonzero |= (w[i] != 0);
        }
        char buf[16];
        if (nonzero) { snprintf(buf, sizeof buf, "%09u", (unsigned)rem); out.insert(0, buf); }
        else { snprintf(buf, sizeof buf, "%u", (unsigned)rem); out.insert(0, buf); break; }
    }
    return out;
}

// ---- randomness ------------------------------------------------------------------------------------------
namespace {
std::mutex g_rand_mutex;
std::vector<uint8_t> g_override;
size_t g_override_pos = 0;
}  // namespace

// The override makes r, s and the round randomness deterministic for every prover of the process, i.e. it removes zero
// knowledge: it only exists in processes started with ULTRAGROTH_TEST_HOOKS=1 (tests, bench.py --check, smoke()).
bool testHooksEnabled() {
    static const bool on = [] { const char* e = getenv("ULTRAGROTH_TEST_HOOKS"); return e && e[0] == '1' && !e[1]; }();
    return on;
}
bool setRandomOverride(const void* bytes, size_t n) {
    if (n && !testHooksEnabled()) return false;               // clearing is always allowed
    std::lock_guard<std::mutex> lock(g_rand_mutex);
    g_override.assign(static_cast<const uint8_t*>(bytes), static_cast<const uint8_t*>(bytes) + n);
    g_override_pos = 0;
    return true;
}
// For the CLIs, which have no call to make: under ULTRAGROTH_TEST_HOOKS=1 the environment variable ULTRAGROTH_TEST_BLINDING
// (hex) sets the override once, before the first draw (tests compare the files `prover` writes byte for byte).
static void overrideFromEnvOnce() {
    static const bool done = [] {
        const char* hex = testHooksEnabled() ? getenv("ULTRAGROTH_TEST_BLINDING") : nullptr;
        if (!hex) return true;
        std::vector<uint8_t> bytes;
        auto nib = [](char ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; };
        for (size_t i = 0; hex[i] && hex[i + 1]; i += 2) {
            const int hi = nib(hex[i]), lo = nib(hex[i + 1]);
            if (hi < 0 || lo < 0) break;
            bytes.push_back((uint8_t)(hi * 16 + lo));
        }
        if (!bytes.empty()) setRandomOverride(bytes.data(), bytes.size());
        return true;
    }();
    (void)done;
}
void randomBytes(void* buf, size_t n) {
    overrideFromEnvOnce();
    {
        std::lock_guard<std::mutex> lock(g_rand_mutex);
        if (!g_override.empty()) {
            uint8_t* o = static_cast<uint8_t*>(buf);
            for (size_t i = 0; i < n; i++) { o[i] = g_override[g_override_pos]; g_override_pos = (g_override_pos + 1) % g_override.size(); }
            return;
        }
    }
    uint8_t* o = static_cast<uint8_t*>(buf);
    size_t done = 0;
    while (done < n) {
        ssize_t r = getrandom(o + done, n - done, 0);
        if (r < 0) { if (errno == EINTR) continue; throw std::system_error(errno, std::generic_category(), "getrandom"); }
        done += (size_t)r;
    }
}

// ---- Keccak-256 ------------------------------------------------------------------------------------------
namespace {
const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
void permute(uint64_t a[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        // rho + pi
        uint64_t b[25];
        int x = 1, y = 0;
        b[0] = a[0];
        uint64_t cur = a[1];
        for (int t = 0; t < 24; t++) {
            int r = ((t + 1) * (t + 2) / 2) % 64;
            int nx = y, ny = (2 * x + 3 * y) % 5;
            uint64_t next = a[nx + 5 * ny];
            b[nx + 5 * ny] = r ? rotl(cur, r) : cur;
            cur = next; x = nx; y = ny;
        }
        // chi
        for (int yy = 0; yy < 25; yy += 5)
            for (int xx = 0; xx < 5; xx++) a[yy + xx] = b[yy + xx] ^ (~b[yy + (xx + 1) % 5] & b[yy + (xx + 2) % 5]);
        a[0] ^= RC[round];
    }
}
}  // namespace

void keccak256(uint8_t out[32], const uint8_t* in, uint64_t len) {
    uint64_t st[25];
    memset(st, 0, sizeof st);
    uint8_t* sb = reinterpret_cast<uint8_t*>(st);
    const uint64_t rate = 136;
    while (len >= rate) {
        for (uint64_t i = 0; i < rate; i++) sb[i] ^= in[i];
        permute(st);
        in += rate; len -= rate;
    }
    for (uint64_t i = 0; i < len; i++) sb[i] ^= in[i];
    sb[len] ^= 0x01;
    sb[rate - 1] ^= 0x80;
    permute(st);
    memcpy(out, sb, 32);
}

}  // namespace ughost
