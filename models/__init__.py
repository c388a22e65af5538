"""Alias package: `models.*` re-exports avsum_amd.models.* so that the reference's scripts, which import
`models.*` and `src.models.*`, run unmodified against the MI355X implementation."""
