// SHA-512 (FIPS 180-4), host side: Ristretto255Group::hash_to_scalar hashes with SHA-512
// (reference src/groups/ristretto255.rs:196-205).
#pragma once
#include <stddef.h>
#include <stdint.h>
namespace mpvss {
void sha512(const void* data, size_t len, uint8_t out[64]);
}
