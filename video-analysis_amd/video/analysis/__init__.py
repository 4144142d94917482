"""video.analysis -- per-frame image operations (labelling, regions, moments, statistics)
of the reference's video/analysis package that lie on the GPU hot path."""
