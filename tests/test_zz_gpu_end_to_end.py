"""GPU tier of the end-to-end cases that were added after the round's GPU budget was spent: the same inputs and goldens as
their emulation-tier twins (tests/test_pipeline.py, tests/test_genotype.py), through the real library.  The kernels they
run are the ones the other GPU tests cover; the file sorts last so that it cannot hide another file's result under `-x`."""
import pytest

import cases
from test_genotype import run_genotype_vcf
from test_pipeline import run_population, run_sample

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(cases.SAMPLES_EMU))
def test_more_samples_bam_to_vcf_and_snf(name, tmp_path):
    run_sample(name, tmp_path, None, through_file=False)


@pytest.mark.parametrize("name", sorted(cases.POPULATIONS))
def test_bams_to_merged_vcf(name, tmp_path):
    run_population(name, tmp_path, None)


@pytest.mark.parametrize("name", ["sample_splits_14x", "sample_two_contigs_12x"])
def test_genotype_vcf_end_to_end(name):
    run_genotype_vcf(name, None)
