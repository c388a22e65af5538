"""Alias package: `data.*` re-exports avsum_amd.data.* so that the reference's scripts, which import
`data.*` and `src.data.*`, run unmodified against the MI355X implementation."""
