// alphabet.h -- symbol <-> {ASCII, index, bit-plane code} maps, host and device.
// Behaviour of /root/reference src/alphabet.rs:169-413: nucleotide indices $0 A1 C2 G3 N4 T5 (U == T,
// any other byte -> N), amino indices $0 A1 C2 D3 E4 F5 G6 H7 I8 K9 L10 M11 N12 P13 Q14 R15 S16 T17 V18
// W19 X20 Y21 (any other byte -> X); case-insensitive (src/alphabet.rs:109-114); '#' == '$'.
#pragma once
#include "layout.h"

namespace awry {

// index -> 5-bit plane code, src/alphabet.rs:280-303
AWRY_HD uint32_t aa_code_of_index(int idx) {
  constexpr uint8_t T[22] = {0x00, 0x0C, 0x17, 0x03, 0x06, 0x1E, 0x1A, 0x1B, 0x19, 0x15, 0x1C,
                             0x1D, 0x08, 0x09, 0x04, 0x13, 0x0A, 0x05, 0x16, 0x01, 0x1F, 0x02};
  return (unsigned)idx < 22u ? T[idx] : 0x1F;
}
// index -> 3-bit plane code, src/alphabet.rs:318-325
AWRY_HD uint32_t nt_code_of_index(int idx) {
  constexpr uint8_t T[6] = {4, 6, 5, 3, 2, 1};
  return (unsigned)idx < 6u ? T[idx] : 2;
}
AWRY_HD uint32_t code_of_index(int alphabet, int idx) {
  return alphabet == NUCLEOTIDE ? nt_code_of_index(idx) : aa_code_of_index(idx);
}
// code -> index, src/alphabet.rs:199-222,237-244 (unused codes decode to the ambiguity symbol)
AWRY_HD int nt_index_of_code(uint32_t code) {
  constexpr int8_t T[8] = {4, 5, 4, 3, 0, 2, 1, 4};
  return T[code & 7];
}
AWRY_HD int aa_index_of_code(uint32_t code) {
  constexpr int8_t T[32] = {0,  19, 21, 3,  14, 17, 4,  20, 12, 13, 16, 20, 1,  20, 20, 20,
                            20, 20, 20, 15, 20, 9,  18, 2,  20, 8,  6,  7,  10, 11, 5,  20};
  return T[code & 31];
}
AWRY_HD int index_of_code(int alphabet, uint32_t code) {
  return alphabet == NUCLEOTIDE ? nt_index_of_code(code) : aa_index_of_code(code);
}

// ASCII byte -> symbol index
AWRY_HD int index_of_ascii(int alphabet, uint8_t a) {
  if (a >= 'a' && a <= 'z') a = (uint8_t)(a - 32);
  if (a == '$' || a == '#') return 0;
  if (alphabet == NUCLEOTIDE) {
    switch (a) {
      case 'A': return 1;
      case 'C': return 2;
      case 'G': return 3;
      case 'T': case 'U': return 5;
      default: return 4;
    }
  }
  switch (a) {
    case 'A': return 1;  case 'C': return 2;  case 'D': return 3;  case 'E': return 4;
    case 'F': return 5;  case 'G': return 6;  case 'H': return 7;  case 'I': return 8;
    case 'K': return 9;  case 'L': return 10; case 'M': return 11; case 'N': return 12;
    case 'P': return 13; case 'Q': return 14; case 'R': return 15; case 'S': return 16;
    case 'T': return 17; case 'V': return 18; case 'W': return 19; case 'Y': return 21;
    default: return 20;
  }
}
AWRY_HD uint8_t ascii_of_index(int alphabet, int idx) {
  if (alphabet == NUCLEOTIDE) {
    constexpr char T[7] = "$ACGNT";
    return (unsigned)idx < 6u ? (uint8_t)T[idx] : (uint8_t)'N';
  }
  constexpr char T[23] = "$ACDEFGHIKLMNPQRSTVWXY";
  return (unsigned)idx < 22u ? (uint8_t)T[idx] : (uint8_t)'X';
}

}  // namespace awry
