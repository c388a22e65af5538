"""Alias package: `src.features.*` re-exports avsum_amd.features.* so that the reference's scripts, which import
`features.*` and `src.features.*`, run unmodified against the MI355X implementation."""
