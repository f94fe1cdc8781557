"""Experiment configurations: plain modules whose attributes are the hyper-parameters
(same attribute surface as the reference's phiseg/experiments/*.py, SURVEY.md section 8(b) surface B3)."""
