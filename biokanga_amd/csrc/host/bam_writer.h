// bam_writer.h - BGZF-compressed BAM + BAI output, byte for byte what the reference's CSAMfile / bgzf write
// (libbiokanga/SAMfile.cpp:1383-2628, libbiokanga/bgzf.cpp:222-500) for the same records:
//   * the uncompressed stream (header, records) is cut into 0xff00-byte blocks; one extra cut follows the last
//     aligned record (the reference calls bgzf_flush there, SAMfile.cpp:2530-2531); each block is deflated on
//     its own (raw deflate, level 6, memLevel 8) behind the 18-byte BGZF header; the 28-byte EOF block ends it
//   * <file>.bai: per reference sequence up to the one holding the last aligned record: the bins that own
//     chunks in bin order, chunks merged while alignments touch (CSAMfile::AddChunk), then the 16 kb linear
//     index with a virtual offset only where an alignment STARTS in the window (zero elsewhere)
// The blocks are independent, so they are compressed by all host threads; virtual offsets follow from the
// prefix sum of the compressed sizes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bk {

struct BamAligned {         // one aligned record, in stream order
    uint64_t u_beg, u_end;  // byte range of the record in the uncompressed stream
    int32_t ref;            // index into the header's reference list
    int32_t pos, end;       // 0-based first and last reference base covered
};

// bin of [beg, end) - the formula of the SAM specification as the reference uses it
int bam_reg2bin(int beg, int end);

// max_ref_len: the longest sequence of the header; from 512 Mbp on the index is a BGZF-compressed CSI (<file>.csi; min_shift 14,
// depth from that length) instead of a BAI, as in the reference (SAMfile.cpp:1602-1607,1664-1695,1751-1762,1870-1875).
// stream: whole uncompressed BAM (magic, header text, references, records).  aligned: the aligned records in
// stream order.  flush_at: stream offset right after the last aligned record (0 when there is none).
// Returns 0 or a negative teBSFrsltCodes value.
int write_bam_and_bai(const std::string &path, const std::vector<uint8_t> &stream, const std::vector<BamAligned> &aligned,
                      uint64_t flush_at, uint32_t n_refs, uint64_t max_ref_len, int nthreads, std::string *err);

}  // namespace bk
