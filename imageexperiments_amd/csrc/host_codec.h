// host_codec.h -- product host code: whole-image helpers around the tile records.
//   decode_image      compressed::decodeImage           CompressedImage.cpp:783-835
//                     matching::FromCoeffsDynamic        MatchingPursuit.cpp:109-128
//                     img::RGBFromYUV                    ImageHelper/src/misc.cpp:28-36
//   psnr              compressed::calculatePSNR          CompressedImage.cpp:343-357
// Decoding is SURVEY 8(f) row N1 ("next"): it runs on the host in this round; the GPU decoder comes later.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "host_bitstream.h"
#include "host_dictionary.h"

namespace mpc {

// rgb_out: width*height*3 bytes, row-major.  false = inconsistent streams.
bool decode_streams(const Dictionary& dict, const Streams& s, std::vector<uint8_t>& rgb_out);

double psnr(const uint8_t* original, const uint8_t* decoded, int width, int height);

}  // namespace mpc
