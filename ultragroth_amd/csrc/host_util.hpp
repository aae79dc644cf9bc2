// host_util.hpp -- host-side pieces shared by the prover classes: iden3 binfile container, zkey / wtns
// headers, decimal printing, blinding randomness, Keccak-256.
//
// Mirrors, with its own code, the behaviour of the reference's
//   BinFileUtils::BinFile          src/binfile_utils.cpp:24-176 (error strings included)
//   ZKeyUtils::loadHeader          src/zkey_utils.cpp:42-76, ultra_groth_loadHeader :123-163
//   WtnsUtils::loadHeader          src/wtns_utils.cpp:13-26
//   BinFileUtils::FileLoader       src/fileloader.cpp:23-59 (mmap, MADV_SEQUENTIAL)
//   randombytes_buf                src/random_generator.hpp:4-25
//   FIPS202_KECCAK_256             src/keccak256.cpp:8 (Keccak-f[1600], rate 1088, padding 0x01)
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ughost {

struct Section { const uint8_t* start; uint64_t size; };

class BinFile {
public:
    BinFile(const void* data, uint64_t size, const std::string& type, uint32_t maxVersion);
    const uint8_t* sectionData(uint32_t id, uint32_t pos = 0) const;
    uint64_t sectionSize(uint32_t id, uint32_t pos = 0) const;
    bool hasSection(uint32_t id) const { return sections_.count(id) != 0; }
private:
    const Section& find(uint32_t id, uint32_t pos) const;
    std::map<uint32_t, std::vector<Section>> sections_;
};

class FileMap {           // read-only mmap of a whole file
public:
    explicit FileMap(const std::string& path);
    ~FileMap();
    const void* data() const { return addr_; }
    uint64_t size() const { return size_; }
    FileMap(const FileMap&) = delete;
    FileMap& operator=(const FileMap&) = delete;
private:
    void* addr_ = nullptr; uint64_t size_ = 0; int fd_ = -1;
};

struct ZkeyHeader {
    uint32_t n8q = 0, n8r = 0, nVars = 0, nPublic = 0, domainSize = 0;
    uint64_t nCoefs = 0;
    bool rIsBn254 = false;
    const uint8_t *alpha1 = nullptr, *beta1 = nullptr, *beta2 = nullptr, *gamma2 = nullptr, *delta1 = nullptr, *delta2 = nullptr;
    // UltraGroth (protocol 1337): delta1/delta2 above hold the FINAL-round deltas
    uint32_t numIndexesC1 = 0, numIndexesC2 = 0, randIndx = 0;
    const uint8_t *roundDelta1 = nullptr, *roundDelta2 = nullptr;
};
ZkeyHeader loadZkeyHeader(const BinFile& f, bool ultra);

struct WtnsHeader { uint32_t n8 = 0, nVars = 0; bool primeIsBn254 = false; };
WtnsHeader loadWtnsHeader(const BinFile& f);

// 32-byte little-endian plain integer -> decimal string
std::string toDecimal(const uint8_t le[32]);

// blinding randomness: OS entropy unless a test override is queued (ug_test_set_blinding)
void randomBytes(void* buf, size_t n);
bool setRandomOverride(const void* bytes, size_t n);      // false (and no effect) unless ULTRAGROTH_TEST_HOOKS=1
bool testHooksEnabled();

void keccak256(uint8_t out[32], const uint8_t* in, uint64_t len);

}  // namespace ughost
