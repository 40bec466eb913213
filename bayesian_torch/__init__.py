"""Alias package: the reference's import paths, served by bayesian_torch_amd (drop-in boundary,
SURVEY.md section 8(b)).  Contains no code of its own."""
