// host_codec.h -- product host code: whole-image helpers around the decoded pixels.
//   psnr              compressed::calculatePSNR          CompressedImage.cpp:343-357
// Tile reconstruction (compressed::decodeImage / matching::FromCoeffsDynamic / img::RGBFromYUV) runs on the
// device only: mp_decode_kernel in mp_kernels.hip behind mpc_decode_image / mpc_decode_tiles_device.
#pragma once
#include <cstddef>
#include <cstdint>

namespace mpc {

double psnr(const uint8_t* original, const uint8_t* decoded, int width, int height);

}  // namespace mpc
