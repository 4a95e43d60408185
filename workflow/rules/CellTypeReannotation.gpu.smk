### MI355X replacement of the SNV branch of LongSom's workflow/rules/CellTypeReannotation.smk (same OUTPUT files).
#
# Option A (fused, recommended): rule Reannotation_gpu runs pass-1 calling -> HCCV -> per-cell genotyping -> re-annotation
# AND the pass-2 SNV calling (rules/SNVCalling.smk) in ONE process: the BAM is decoded once, the reads stay in HBM, pass 2 only
# swaps the barcode -> cell-type table.  Option B: the per-rule drop-ins below (same rule names as the reference with a
# _gpu suffix) keep the reference's rule graph; every script under scripts_gpu/CellTypeReannotation/ has the flag surface
# of the script it replaces.  The fusion branch (ctat-LR-fusion, HCCVFusions) is unchanged and optional here.

GPU_SCRIPTS = str(workflow.basedir) + "/scripts_gpu"

rule Reannotation_gpu:
    input:
        bam=f"{INPUT}/bam/{{id}}.bam",
        barcodes="Barcodes/{id}.tsv",
        ref=str(workflow.basedir)+config['Reference']['genome'],
        pon_LR="PoN/PoN/PoN_LR.tsv" if PON else [],
        pon_SR=str(workflow.basedir)+config['Reference']['PoN_SR'],
        RNA_editing=str(workflow.basedir)+config['Reference']['RNA_editing'],
        fusions="CellTypeReannotation/HCCV/{id}.Fusions.SingleCellGenotype.tsv",
    output:
        hccv="CellTypeReannotation/HCCV/{id}.HCCV.tsv",
        genotype="CellTypeReannotation/HCCV/{id}.SNVs.SingleCellGenotype.tsv",
        barcodes="CellTypeReannotation/ReannotatedCellTypes/{id}.tsv",
        step3="SNVCalling/BaseCellCalling/{id}.calling.step3.tsv",
    params:
        script=GPU_SCRIPTS+"/CellTypeReannotation/longsom_gpu_reannotation.py",
        gnomAD_db=str(workflow.basedir)+config['Reference']['gnomAD_db'],
        gz_compat="--p1_reference_gz_compat --p2_reference_gz_compat" if config['Run'].get('reference_gz_compat', False) else "",
        # Run.htslib_legacy_del_merge: True counts CIGAR 1D2D's first deleted column as 'D' (pysam over htslib <= 1.10); default: htslib >= 1.11
        htslib="--htslib_legacy_del_merge" if config['Run'].get('htslib_legacy_del_merge', False) else "",
        # Run.allow_missing_gnomad: True runs step 2 without its gnomAD filter when the database cannot be read (default: the rule fails, as the reference's gnomAD_DB() does)
        no_gnomad="--allow_missing_gnomad" if config['Run'].get('allow_missing_gnomad', False) else "",
        # pass 1 = config['Reanno'], pass 2 = config['SNVCalling'].  Pass 1 deliberately gets NO min_ac_reads / min_ac_cells: the
        # reference's pass-1 step-1 rule does not forward them either (rules/CellTypeReannotation.smk:208-238)
        r1=config['Reanno']['BaseCellCalling'],
        m1=config['Reanno']['BaseCellCounter']['min_mapping_quality'],
        h=config['Reanno']['HCCV'],
        r=config['Reanno']['Reannotation'],
        c2=config['SNVCalling']['BaseCellCalling'],
        m2=config['SNVCalling']['BaseCellCounter']['min_mapping_quality'],
        # GPUs of this node used by the rule: one rank per GPU, every rank keeps its region's reads in HBM across both passes
        launcher=lambda wc, resources: "python" if resources.gpu == 1 else f"python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node {resources.gpu}",
    resources:
        gpu=config['Run'].get('gpus', 1)
    log:
        "logs/Reannotation_gpu/{id}.log",
    shell:
        r"""
        {params.launcher} {params.script} --bam {input.bam} --meta {input.barcodes} --ref {input.ref} --id {wildcards.id} --outdir . \
        --fusions {input.fusions} --editing {input.RNA_editing} --pon_SR {input.pon_SR} --pon_LR {input.pon_LR} --gnomAD_db {params.gnomAD_db} {params.gz_compat} {params.htslib} {params.no_gnomad} \
        --p1_min_mapping_quality {params.m1} --p1_min_cell_types {params.r1[Min_cell_types]} --p1_min_distance {params.r1[min_distance]} \
        --p1_max_gnomad_vaf {params.r1[max_gnomAD_VAF]} \
        --p1_alpha1 {params.r1[alpha1]} --p1_beta1 {params.r1[beta1]} --p1_alpha2 {params.r1[alpha2]} --p1_beta2 {params.r1[beta2]} \
        --reanno_hccv_min_depth {params.h[min_depth]} --reanno_hccv_delta_vaf {params.h[deltaVAF]} --reanno_hccv_delta_mcf {params.h[deltaMCF]} \
        --reanno_hccv_clust_dist {params.h[clust_dist]} --reanno_chrm_contaminant {params.h[chrM_contaminant]} --reanno_alt_flag {params.h[alt_flag]} \
        --reanno_pvalue {params.h[pvalue]} --reanno_min_variants {params.r[min_variants]} --reanno_min_fraction {params.r[min_fraction]} \
        --p2_min_mapping_quality {params.m2} --p2_min_cell_types {params.c2[Min_cell_types]} --p2_min_distance {params.c2[min_distance]} \
        --p2_max_gnomad_vaf {params.c2[max_gnomAD_VAF]} --p2_delta_vaf {params.c2[deltaVAF]} --p2_delta_mcf {params.c2[deltaMCF]} \
        --p2_min_ac_reads {params.c2[min_ac_reads]} --p2_min_ac_cells {params.c2[min_ac_cells]} --p2_clust_dist {params.c2[clust_dist]} \
        --p2_alpha1 {params.c2[alpha1]} --p2_beta1 {params.c2[beta1]} --p2_alpha2 {params.c2[alpha2]} --p2_beta2 {params.c2[beta2]} > {log}
        """

rule HighConfidenceCancerVariants_gpu:
    input:
        tsv="CellTypeReannotation/BaseCellCalling/{id}.calling.step2.tsv"
    output:
        tsv="CellTypeReannotation/HCCV/{id}.HCCV.tsv"
    params:
        script=GPU_SCRIPTS+"/CellTypeReannotation/HighConfidenceCancerVariants.py",
        h=config['Reanno']['HCCV'],
    shell:
        "python {params.script} --SNVs {input.tsv} --outfile CellTypeReannotation/HCCV/{wildcards.id} --min_dp {params.h[min_depth]} "
        "--deltaVAF {params.h[deltaVAF]} --deltaMCF {params.h[deltaMCF]} --clust_dist {params.h[clust_dist]}"

rule HCCVSingleCellGenotype_gpu:
    input:
        tsv="CellTypeReannotation/HCCV/{id}.HCCV.tsv",
        bam=f"{INPUT}/bam/{{id}}.bam",
        barcodes="Barcodes/{id}.tsv",
        ref=str(workflow.basedir)+config['Reference']['genome'],
    output:
        tsv="CellTypeReannotation/HCCV/{id}.SNVs.SingleCellGenotype.tsv",
        tmp=temp(directory("CellTypeReannotation/HCCV/{id}/"))
    params:
        script=GPU_SCRIPTS+"/CellTypeReannotation/HCCVSingleCellGenotype.py",
        h=config['Reanno']['HCCV'],
        c=config['Reanno']['BaseCellCalling'],
        mapq=config['Reanno']['BaseCellCounter']['min_mapping_quality'],
    resources:
        gpu=1
    shell:
        "python {params.script} --bam {input.bam} --infile {input.tsv} --ref {input.ref} --outfile {output.tsv} --meta {input.barcodes} "
        "--alt_flag {params.h[alt_flag]} --nprocs {threads} --min_mq {params.mapq} --pvalue {params.h[pvalue]} --alpha2 {params.c[alpha2]} "
        "--beta2 {params.c[beta2]} --chrM_contaminant {params.h[chrM_contaminant]} --tmp_dir {output.tmp}"

rule CellTypeReannotation_gpu:
    input:
        SNVs="CellTypeReannotation/HCCV/{id}.SNVs.SingleCellGenotype.tsv",
        fusions="CellTypeReannotation/HCCV/{id}.Fusions.SingleCellGenotype.tsv",
        barcodes="Barcodes/{id}.tsv"
    output:
        barcodes="CellTypeReannotation/ReannotatedCellTypes/{id}.tsv"
    params:
        script=GPU_SCRIPTS+"/CellTypeReannotation/CellTypeReannotation.py",
        r=config['Reanno']['Reannotation'],
    shell:
        "python {params.script} --SNVs {input.SNVs} --fusions {input.fusions} --outfile {output.barcodes} --meta {input.barcodes} "
        "--min_variants {params.r[min_variants]} --min_frac {params.r[min_fraction]}"
