"""File-name suffixes of the bitstream working directory (src/real_life/utils.py:10-13)."""
BITSTREAM_SUFFIX = ''
GOP_HEADER_SUFFIX = 'h'
GOP_SUFFIX = 'g'
VIDEO_HEADER_SUFFIX = 'v'
