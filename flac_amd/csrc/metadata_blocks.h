/* Internal to the host shims: byte image of one metadata block. */
#ifndef FLACAMD_METADATA_BLOCKS_H
#define FLACAMD_METADATA_BLOCKS_H
#include "../../include/flac_amd_stream.h"

/* Allocates *out (free() it) holding the 4-byte block header and the block data exactly as the
 * reference writes them (stream_encoder_framing.c:50-217); `vendor` replaces a VORBIS_COMMENT's
 * vendor string.  Returns 0, or -1 when the block is too long or memory ran out. */
int flacamd_metadata_block_image(const FLAC__StreamMetadata *m, const char *vendor, uint8_t **out, size_t *out_len);

/* Build the object the decoder hands to the metadata callback from the `n` data bytes of a block of the
 * given type (not STREAMINFO).  Returns 0, or -1 for truncated data / no memory (nothing is left allocated).
 * flacamd_metadata_block_free releases what parse allocated. */
int flacamd_metadata_block_parse(unsigned type, int is_last, const uint8_t *data, size_t n, FLAC__StreamMetadata *m);
void flacamd_metadata_block_free(FLAC__StreamMetadata *m);
#endif
