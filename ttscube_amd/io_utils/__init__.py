"""Mirror of the reference's ``cube.io_utils`` pieces that sit directly on either side of the hot path."""
